"""Exponential moving average of the U-Net weights - mirrors /root/reference diffusion/algorithms/ema.py
(``EMA`` :88-370, ``compute_ema`` :26-76; configured at yamls/hydra-yamls/SD-2-base-512.yaml:8-13 with
``smoothing: 0.9999, update_interval: 1ba, ema_start: 800000ba``).

Here the EMA state is one flat fp32 buffer and the update ``ema = s*ema + (1-s)*w`` is fused into the AdamW kernel
(one extra read-modify-write stream over the parameters instead of a per-tensor Python loop).  ``swap_params`` exchanges
live and EMA weights (the reference does this around eval / checkpointing, ema.py:243-278)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from ..trainer import Callback, _parse_time


class EMA(Callback):

    def __init__(self, half_life: Optional[str] = '1000ba', smoothing: Optional[float] = None, ema_start: str = '0ba',
                 update_interval: Optional[str] = None):
        if half_life is None and smoothing is None:
            raise ValueError('Either half_life or smoothing must be specified')
        if half_life is not None and smoothing is not None:
            raise ValueError('Only one of  half_life or smoothing can be specified')
        self.update_interval = _parse_time(update_interval)[0] if update_interval else 1
        if smoothing is None:
            hl = _parse_time(half_life)[0] / self.update_interval
            smoothing = math.exp(-math.log(2) / hl)
        self.smoothing = float(smoothing)
        start, unit = _parse_time(ema_start)
        self.ema_start_batches = None if unit == 'dur' else int(start)
        self.ema_start_frac = float(start) if unit == 'dur' else None
        self.ema_started = False
        self.ema_weights_active = False

    def _start_batch(self, trainer) -> int:
        if self.ema_start_batches is not None:
            return self.ema_start_batches
        return int(self.ema_start_frac * trainer.max_batches)

    def before_optimizer_step(self, trainer):
        """Called by the trainer right before ``optimizer.step()`` of batch ``trainer.batch_idx`` (0-based)."""
        opt = trainer.optimizer
        done = trainer.batch_idx + 1
        if (not self.ema_started and done > self._start_batch(trainer)) or (self.ema_started and opt.ema is None):
            opt.ema = trainer.model.unet.master.clone()  # start the average from the current weights
            opt.ema_smoothing = self.smoothing
            self.ema_started = True
        opt.ema_update_this_step = self.ema_started and (done % self.update_interval == 0)

    # checkpoint state (reference: ema.py:280-336 serialises the shadow parameters and its bookkeeping; here the shadow
    # itself lives in the optimizer's flat buffers and is saved there)
    def state_dict(self):
        return {'smoothing': self.smoothing, 'ema_started': self.ema_started, 'ema_weights_active': self.ema_weights_active,
                'update_interval': self.update_interval}

    def load_state_dict(self, sd):
        self.ema_started = bool(sd['ema_started'])
        self.ema_weights_active = bool(sd.get('ema_weights_active', False))
        self.smoothing = float(sd.get('smoothing', self.smoothing))

    def swap_params(self, trainer):
        """Exchange live and averaged weights (call again to swap back)."""
        opt, unet = trainer.optimizer, trainer.model.unet
        if opt.ema is None:
            return
        tmp = unet.master.clone()
        unet.master.copy_(opt.ema)
        opt.ema.copy_(tmp)
        unet.sync_shadows()
        self.ema_weights_active = not self.ema_weights_active
