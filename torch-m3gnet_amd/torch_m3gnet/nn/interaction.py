"""Import-path alias (reference: nn/interaction.py); implementation in nn/modules.py."""
from ._bessel_zeros import SPHERICAL_BESSEL_ZEROS  # noqa: F401
from .modules import (  # noqa: F401
    NormalizedSphericalBessel, ThreeBodyInteration, cutoff_function, legendre_cos, spherical_bessel,
)
