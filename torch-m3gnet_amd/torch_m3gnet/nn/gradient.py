"""Import-path alias (reference: nn/gradient.py); implementation in nn/modules.py."""
from .modules import Gradient  # noqa: F401
