"""Sampling side of the diffusion package: sigma generators for training (sigma_generators.py), the sampler classes
(sampling.py) and the MI355X denoiser they drive (fused.py).  Each module lists its public names in __all__."""
from . import fused as _fused, sampling as _sampling, sigma_generators as _sigma_generators
from .fused import *  # noqa: F401,F403
from .sampling import *  # noqa: F401,F403
from .sigma_generators import *  # noqa: F401,F403

__all__ = sorted(_fused.__all__ + _sampling.__all__ + _sigma_generators.__all__)
