"""Import-path alias (reference: nn/scale.py); implementation in nn/modules.py."""
from .modules import ScaleLength  # noqa: F401
