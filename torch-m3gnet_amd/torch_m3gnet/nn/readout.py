"""Import-path alias (reference: nn/readout.py); implementation in nn/modules.py."""
from .modules import AtomWiseReadout  # noqa: F401
