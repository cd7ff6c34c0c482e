from .embedding import AbstractEmbModel, GeneralConditioner, PrecomputedEmbedder
from .metadata import ConcatTimestepEmbedderND

__all__ = ["AbstractEmbModel", "GeneralConditioner", "PrecomputedEmbedder", "ConcatTimestepEmbedderND"]
