"""Fused AdamW over the U-Net's flat buffers (one HIP launch for 866 M parameters).

Stands where diffusion/train.py:33 instantiates ``torch.optim.AdamW`` from yamls/hydra-yamls/SD-2-base-256.yaml:55-58
(lr 1e-4, weight_decay 0.01, torch-default betas/eps).  Same update rule as torch.optim.AdamW; additionally writes
the bf16 compute shadow and refreshes the transposed (dgrad) shadow."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):

    def __init__(self, params=None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, unet=None):
        if unet is None:
            raise ValueError('FusedAdamW needs the UNetHIP that owns the flat parameter buffers (unet=...)')
        self.unet = unet
        # torch.optim bookkeeping (param_groups / state_dict) over a single flat tensor
        super().__init__([unet.master], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self.ema = None            # flat fp32 EMA of the weights (set by algorithms.ema.EMA)
        self.ema_smoothing = 0.0
        self.ema_update_this_step = False

    # The update can be issued in slices as gradients become final (trainer + parallel.BucketedAllReducer hand over
    # [lo, hi) ranges back-to-front during the last microbatch's backward, on the reducer's side stream), so the
    # HBM-bound optimizer pass hides under the remaining backward GEMMs instead of trailing the step.
    @torch.no_grad()
    def begin_step(self):
        """Open optimizer step number opt_step+1; nothing updated yet."""
        self.unet.opt_step += 1
        self._pending_hi = self.unet.master.numel()
        self._open = True

    @torch.no_grad()
    def step_range(self, lo: int, hi: int):
        """AdamW (+ bf16 shadow, + EMA) on flat offsets [lo, hi) of the step opened by begin_step()."""
        if hi <= lo:
            return
        g = self.param_groups[0]
        u = self.unet
        ema = self.ema[lo:hi] if (self.ema is not None and self.ema_update_this_step) else None
        ops.adamw(u.master[lo:hi], u.grad[lo:hi], u.exp_avg[lo:hi], u.exp_avg_sq[lo:hi], u.shadow[lo:hi], g['lr'],
                  g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], u.opt_step, self.grad_scale,
                  ema=ema, ema_smoothing=self.ema_smoothing)
        self._pending_hi = min(self._pending_hi, lo)

    @torch.no_grad()
    def step(self, closure=None):
        """Update everything begin_step()/step_range() have not covered yet, then refresh the dgrad weight shadow."""
        if not getattr(self, '_open', False):
            self.begin_step()
        self.step_range(0, self._pending_hi)
        self._open = False
        self.unet.refresh_transposed()

    def zero_grad(self, set_to_none: bool = False):
        self.unet.grad.zero_()

    def state_dict(self):
        """CPU copies (a checkpoint must not alias live device buffers that the next step overwrites)."""
        u = self.unet
        sd = {'step': u.opt_step, 'exp_avg': u.exp_avg.detach().cpu().clone(), 'exp_avg_sq': u.exp_avg_sq.detach().cpu().clone(),
              'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}
        if self.ema is not None:
            sd['ema'] = self.ema.detach().cpu().clone()
            sd['ema_smoothing'] = self.ema_smoothing
        return sd

    def load_state_dict(self, sd):
        u = self.unet
        u.opt_step = int(sd['step'])
        u.exp_avg.copy_(sd['exp_avg'])
        u.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for k, v in sd['param_groups'][0].items():
            self.param_groups[0][k] = v
        if 'ema' in sd:
            self.ema = sd['ema'].to(device=u.master.device, dtype=u.master.dtype).clone()
            self.ema_smoothing = float(sd['ema_smoothing'])
