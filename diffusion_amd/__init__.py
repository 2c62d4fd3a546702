"""diffusion_amd: MI355X-native Stable Diffusion 2 U-Net training step (HIP kernels behind the reference's
ComposerModel / dataloader surfaces).  See DESIGN.md."""
__version__ = '0.1.0'
