"""MI355X-native drop-in for the compute core of
SonyCSLParis/interactive-spectrogram-inpainting.

Same module paths and class names as the reference package
(`interactive_spectrogram_inpainting.vqvae.vqvae.VQVAE`, ...); the arithmetic
runs in hand-written gfx950 HIP kernels behind the C-ABI of
`include/isi_hip.h` (loaded by `._hip`).  There is no CPU or PyTorch fallback:
calling a compute method without the built library or on a non-GPU tensor
raises.
"""
__all__ = ["vqvae", "priors", "utils"]
