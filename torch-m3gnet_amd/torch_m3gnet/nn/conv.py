"""Import-path alias (reference: nn/conv.py); implementation in nn/modules.py."""
from .modules import M3GNetConv  # noqa: F401
