"""paddlemix_amd — MI355X (gfx950) native implementation of the ppdiffusers Stable-Diffusion denoising hot path.

Host side mirrors the reference interfaces for this path (UNet2DConditionModel callable, AttnProcessor /
scaled_dot_product_attention_ seams, scheduler API); all arithmetic runs in hand-written HIP kernels behind the
C ABI of ``libmi355x_sd.so`` (include/mi355x_sd.h).  There is no CPU or PyTorch fallback: without the built
library every op raises.
"""
__version__ = "0.1.0"
