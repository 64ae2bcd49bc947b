"""Import-path alias (reference: nn/core.py); implementation in nn/modules.py."""
from .modules import GatedMLP  # noqa: F401
