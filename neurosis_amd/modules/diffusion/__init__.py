"""MI355X mirror of neurosis.modules.diffusion (the names the SD/SDXL training configs reference)."""
from .denoiser import Denoiser, DiscreteDenoiser
from .denoiser_preconditioning import (DenoiserPreconditioning, EDMPreconditioning, EpsPreconditioning, RectifiedFlowComfyPreconditioning,
                                        RectifiedFlowXLPreconditioning, VPreconditioning, VPreconditioningWithEDMcNoise)
from .denoiser_weighting import (DenoiserWeighting, EDMWeighting, EpsWeighting, MinSNRGammaModifier, RectifiedFlowComfyWeighting, RectifiedFlowWeighting,
                                 UnitWeighting)
from .discretization import (Discretization, EDMcDiscretization, EDMcSimpleDiscretization, EDMDiscretization, LegacyDDPMDiscretization,
                             RectifiedFlowComfyDiscretization, RectifiedFlowDiscretization, TanZeroSNRDiscretization)
from .loss import DiffusionLoss, StandardDiffusionLoss
from .model import AttnBlock, Decoder, Encoder, MemoryEfficientAttnBlock, ResnetBlock
from .openaimodel import Timestep, UNetModel
from .sampling import (CosineScheduleSigmaGenerator, DiscreteSigmaGenerator, EDMSigmaGenerator, InjectedSigmaGenerator, RectifiedFlowComfySigmaGenerator,
                       RectifiedFlowSigmaGenerator, SigmaGenerator, TanScheduleSigmaGenerator)
from .wrappers import IdentityWrapper, OpenAIWrapper

__all__ = [
    "AttnBlock", "Decoder", "Denoiser", "DenoiserPreconditioning", "DenoiserWeighting", "DiffusionLoss", "DiscreteDenoiser", "DiscreteSigmaGenerator",
    "Discretization", "EDMcDiscretization", "EDMPreconditioning", "EDMSigmaGenerator", "EDMWeighting", "Encoder", "EpsPreconditioning",
    "EpsWeighting", "IdentityWrapper", "InjectedSigmaGenerator", "LegacyDDPMDiscretization", "MemoryEfficientAttnBlock", "OpenAIWrapper",
    "ResnetBlock", "SigmaGenerator", "StandardDiffusionLoss", "Timestep", "UnitWeighting", "UNetModel", "VPreconditioning",
    "CosineScheduleSigmaGenerator", "EDMcSimpleDiscretization", "EDMDiscretization", "MinSNRGammaModifier", "RectifiedFlowComfyDiscretization",
    "RectifiedFlowComfyPreconditioning", "RectifiedFlowComfySigmaGenerator", "RectifiedFlowComfyWeighting", "RectifiedFlowDiscretization",
    "RectifiedFlowSigmaGenerator", "RectifiedFlowWeighting", "RectifiedFlowXLPreconditioning", "TanScheduleSigmaGenerator", "TanZeroSNRDiscretization",
    "VPreconditioningWithEDMcNoise",
]
