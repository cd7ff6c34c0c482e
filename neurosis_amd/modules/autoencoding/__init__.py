"""`neurosis.modules.autoencoding` on MI355X: the loss classes an autoencoder-training config names (losses/)."""
from . import losses

__all__ = ["losses"]
