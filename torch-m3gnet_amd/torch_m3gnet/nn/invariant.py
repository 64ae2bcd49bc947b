"""Import-path alias (reference: nn/invariant.py); implementation in nn/modules.py."""
from .modules import DistanceAndAngle  # noqa: F401
