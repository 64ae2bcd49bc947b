"""Import-path alias (reference: nn/atom_ref.py); implementation in nn/modules.py."""
from .modules import AtomRef  # noqa: F401
