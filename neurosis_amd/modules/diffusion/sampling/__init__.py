from .fused import FusedDenoiser
from .sampling import (
    AncestralSampler,
    BaseDiffusionSampler,
    DPMPP2MSampler,
    DPMPP2SAncestralSampler,
    EDMSampler,
    EulerAncestralSampler,
    EulerEDMSampler,
    HeunEDMSampler,
    LinearMultistepSampler,
    SingleStepDiffusionSampler,
)
from .sigma_generators import DiscreteSigmaGenerator, EDMSigmaGenerator, InjectedSigmaGenerator, SigmaGenerator

__all__ = [
    "AncestralSampler", "BaseDiffusionSampler", "DPMPP2MSampler", "DPMPP2SAncestralSampler", "DiscreteSigmaGenerator", "EDMSampler",
    "EDMSigmaGenerator", "EulerAncestralSampler", "EulerEDMSampler", "FusedDenoiser", "HeunEDMSampler", "InjectedSigmaGenerator",
    "LinearMultistepSampler", "SigmaGenerator", "SingleStepDiffusionSampler",
]
