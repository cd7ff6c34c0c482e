"""MI355X mirror of neurosis.modules.diffusion (the names the SD/SDXL training configs reference)."""
from .denoiser import Denoiser, DiscreteDenoiser
from .denoiser_preconditioning import DenoiserPreconditioning, EDMPreconditioning, EpsPreconditioning, VPreconditioning
from .denoiser_weighting import DenoiserWeighting, EDMWeighting, EpsWeighting, UnitWeighting
from .discretization import Discretization, EDMcDiscretization, LegacyDDPMDiscretization
from .loss import DiffusionLoss, StandardDiffusionLoss
from .model import AttnBlock, Decoder, Encoder, MemoryEfficientAttnBlock, ResnetBlock
from .openaimodel import Timestep, UNetModel
from .sampling import DiscreteSigmaGenerator, EDMSigmaGenerator, InjectedSigmaGenerator, SigmaGenerator
from .wrappers import IdentityWrapper, OpenAIWrapper

__all__ = [
    "AttnBlock", "Decoder", "Denoiser", "DenoiserPreconditioning", "DenoiserWeighting", "DiffusionLoss", "DiscreteDenoiser", "DiscreteSigmaGenerator",
    "Discretization", "EDMcDiscretization", "EDMPreconditioning", "EDMSigmaGenerator", "EDMWeighting", "Encoder", "EpsPreconditioning",
    "EpsWeighting", "IdentityWrapper", "InjectedSigmaGenerator", "LegacyDDPMDiscretization", "MemoryEfficientAttnBlock", "OpenAIWrapper",
    "ResnetBlock", "SigmaGenerator", "StandardDiffusionLoss", "Timestep", "UnitWeighting", "UNetModel", "VPreconditioning",
]
