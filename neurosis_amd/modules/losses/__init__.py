"""Loss-side networks of the autoencoder training path: the PatchGAN discriminator and the discriminator
losses, and the LPIPS perceptual distance over a VGG16 trunk."""
from .functions import HingeDiscLoss, VanillaDiscLoss, get_discr_loss_fn
from .patchgan import NLayerDiscriminator, weights_init
from .perceptual import LPIPS

__all__ = ["HingeDiscLoss", "LPIPS", "NLayerDiscriminator", "VanillaDiscLoss", "get_discr_loss_fn", "weights_init"]
