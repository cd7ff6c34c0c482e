"""Loss-side networks of the autoencoder training path: the PatchGAN discriminator and the discriminator
losses.  LPIPS (torchvision trunks with downloaded weights) is not built."""
from .functions import HingeDiscLoss, VanillaDiscLoss, get_discr_loss_fn
from .patchgan import NLayerDiscriminator, weights_init

__all__ = ["HingeDiscLoss", "NLayerDiscriminator", "VanillaDiscLoss", "get_discr_loss_fn", "weights_init"]
