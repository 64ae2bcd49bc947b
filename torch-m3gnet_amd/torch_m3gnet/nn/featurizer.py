"""Import-path alias (reference: nn/featurizer.py); implementation in nn/modules.py."""
from .modules import AtomFeaturizer, EdgeFeaturizer, EdgeAdjustor  # noqa: F401
