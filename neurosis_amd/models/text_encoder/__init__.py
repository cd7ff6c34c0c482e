"""Frozen text encoders of the SDXL conditioner on MI355X: names of `neurosis.models.text_encoder`.
T5 / image embedders are not on the SD/SDXL path and are not built."""
from .clip import CLIPTextTower, FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2, OpenCLIPTextTower

__all__ = ["CLIPTextTower", "FrozenCLIPEmbedder", "FrozenOpenCLIPEmbedder2", "OpenCLIPTextTower"]
