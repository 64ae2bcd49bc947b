from .modules import (  # noqa: F401
    AtomFeaturizer, AtomRef, AtomWiseReadout, DistanceAndAngle, EdgeAdjustor, EdgeFeaturizer, GatedMLP, Gradient,
    M3GNetConv, NormalizedSphericalBessel, ScaleLength, ThreeBodyInteration, cutoff_function, legendre_cos,
    spherical_bessel,
)
from ._bessel_zeros import SPHERICAL_BESSEL_ZEROS  # noqa: F401
