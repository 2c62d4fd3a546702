"""diffusion_amd: MI355X-native Stable Diffusion 2 U-Net training step (HIP kernels behind the reference's
ComposerModel / dataloader surfaces).  See DESIGN.md."""
import os as _os

# Multi-process GPU work on this pool needs dmabuf IPC: with the legacy mode RCCL and cross-process tensor sharing fail
# with `hipIpcGetMemHandle: invalid argument`.  The HIP runtime reads the variable when it starts, so it is set HERE -
# the one place run.py, bench.py, the tests' worker processes and any `import diffusion_amd...` pass through before the
# first GPU call (importing torch does not start the runtime; torch.cuda.* does).
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

__version__ = '0.1.0'
