"""MI355X-native CycleGAN train-step hot path (gfx950 HIP kernels behind a C ABI)."""
from . import lib, ops  # noqa: F401
from .networks import Discriminator, Generator  # noqa: F401
from .cyclegan import CycleGAN  # noqa: F401
from .schedule import ImagePool, linear_decay_scale  # noqa: F401
from .pipeline import DeviceAugment, UnpairedFolders, UnpairedLoader  # noqa: F401
from .inference import Translator  # noqa: F401

__all__ = ["Generator", "Discriminator", "CycleGAN", "lib"]
