"""videomv_amd — MI355X-native (gfx950) implementation of VideoMV's DDIM denoising hot path.

Importing the package registers the drop-in plugins under the reference's registry names
(``MODEL: UNetSD_T2VBase``, ``DIFFUSION: DiffusionDDIM``); the HIP shared library is loaded on first use and its
absence is a hard error (no CPU fallback)."""
from .registry import (AUTO_ENCODER, DATASETS, DIFFUSION, DISTRIBUTION, EMBEDDER, ENGINE, INFER_ENGINE, MODEL, PRETRAIN,
                       VISUAL, Registry, build_from_config)
from .unet_t2v import UNetSD_T2VBase
from .unet_i2vgen import UNetSD_I2VGen
from .diffusion_ddim import DiffusionDDIM
from .autoencoder import AutoencoderKL

__all__ = ["UNetSD_T2VBase", "UNetSD_I2VGen", "DiffusionDDIM", "AutoencoderKL", "MODEL", "DIFFUSION", "AUTO_ENCODER", "INFER_ENGINE", "Registry"]
