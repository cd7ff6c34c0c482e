"""MI355X mirror of `neurosis.models` (/root/reference/src/neurosis/models/__init__.py:1-17): the names the example
configs reach through this package -- `neurosis.models.DiffusionEngine` (configs/sdxl/sdxl.example.yaml:56,
configs/sd15/sd15.example.yml:56), `neurosis.models.autoencoder.AutoencoderKL`, the frozen CLIP text embedders.

Names of the reference's list that are outside the SD/SDXL training path (T5 / ByT5 / image embedders, the diffusers
autoencoder, the inference wrapper) are not built; asking for one raises an ImportError that says so instead of an
AttributeError that looks like a typo."""
from .autoencoder import AutoencoderKL, AutoencodingEngine
from .diffusion import DiffusionEngine
from .text_encoder import FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2

__all__ = ["AutoencoderKL", "AutoencodingEngine", "DiffusionEngine", "FrozenCLIPEmbedder", "FrozenOpenCLIPEmbedder2"]

_NOT_BUILT = ("AbstractAutoencoder", "AutoencoderKLInferenceWrapper", "IdentityFirstStage", "DiffusersAutoencodingEngine", "FrozenByT5Embedder",
              "FrozenCLIPT5Encoder", "FrozenOpenCLIPImageEmbedder", "FrozenT5Embedder")


def __getattr__(name: str):
    if name in _NOT_BUILT:
        raise ImportError(f"neurosis_amd.models.{name}: not on the SD/SDXL training path (SURVEY.md section 8); this package does not build it")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
