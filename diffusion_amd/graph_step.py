"""hipGraph replay of one microbatch of the U-Net training step (whole-step entry).

One microbatch = noising + U-Net forward + fused MSE loss/gradient + U-Net backward into the flat fp32 gradient
buffer: ~1,700 kernel launches issued from Python through the C ABI.  At the reference YAML's microbatch of 16
(``device_train_microbatch_size: 16``, SD-2-base-256.yaml:87) the kernels are short and the step is bound by the host
issuing them (368 images/s against 1,390 at microbatch 256).  Every C-ABI entry point is capture-safe (no allocation,
no host synchronisation, work only on the given stream), so the whole launch sequence of a microbatch is captured ONCE
per (batch, latent side, dtype, weight) into a hipGraph and replayed: inputs are copied into static buffers, the
activations of the captured microbatch live in the graph's private memory pool, gradients accumulate into the same
flat buffer as in eager mode.

What stays outside the graph: the RNG draws (timesteps, noise), metric updates, the optimizer step and the bucketed
RCCL exchange (with more than one rank the LAST microbatch of a step runs eagerly so that the all-reduce of finished
gradient buckets still overlaps its backward).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops


class GraphedMicrobatch:
    """Captured forward + loss + backward for one input signature."""

    def __init__(self, model, latents: torch.Tensor, cond: torch.Tensor, weight: float):
        unet = model.unet
        dev = unet.device_
        self.model = model
        self.B, _, self.S, _ = latents.shape
        self.weight = float(weight)
        self.lat = torch.empty_like(latents, device=dev)
        self.cond = torch.empty_like(cond, device=dev)
        self.t = torch.zeros(self.B, device=dev, dtype=torch.int64)
        self.noise = torch.empty(latents.shape, device=dev, dtype=torch.float32)
        self.lat.copy_(latents)
        self.cond.copy_(cond)
        self.noise.normal_()
        self.graph = torch.cuda.CUDAGraph()
        # warm-up on a side stream (lazy one-time work - scratch buffers, split-K workspace, hipFuncSetAttribute, the
        # transposed-weight descriptor table - must not happen inside the capture); its gradient contribution is undone
        keep = unet.grad.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._body()
        torch.cuda.current_stream().wait_stream(s)
        unet.grad.copy_(keep)
        del keep
        # buffers the captured launches point at but that live outside the graph's pool: keep them alive with the graph
        # (UNetHIP replaces its scratch when the (batch, side) key changes; SPLITK_WS is module state)
        self._keep = (unet._scratch, unet._ss, unet._coef, unet._delta, ops.SPLITK_WS, unet._tdesc if hasattr(unet, '_tdesc') else None)
        cb, unet._grad_ready_cb = unet._grad_ready_cb, None   # host-side hooks do not belong in a captured sequence
        try:
            with torch.cuda.graph(self.graph):
                self.pred, self.target, self.loss = self._body()
        finally:
            unet._grad_ready_cb = cb

    def _body(self):
        model = self.model
        batch = {model.image_latents_key: self.lat, model.text_latents_key: self.cond}
        out = model.forward(batch, timesteps=self.t, noise=self.noise)
        loss = model.loss(out, batch, weight=self.weight)
        model.backward_from_loss()
        return out[0], out[1], loss.detach()

    def replay(self, latents, cond, timesteps, noise):
        self.lat.copy_(latents, non_blocking=True)
        self.cond.copy_(cond, non_blocking=True)
        self.t.copy_(timesteps, non_blocking=True)
        self.noise.copy_(noise, non_blocking=True)
        self.graph.replay()
        return (self.pred, self.target, self.t), self.loss


class GraphStepCache:
    """Per-model cache of captured microbatches, keyed by the input signature."""

    def __init__(self, model, max_graphs: int = 4):
        self.model = model
        self.max_graphs = max_graphs
        self.graphs: Dict[Tuple, GraphedMicrobatch] = {}

    def usable(self, batch) -> bool:
        m = self.model
        return (ops.PROFILE is None and m.precomputed_latents and m.image_latents_key in batch
                and m.text_latents_key in batch and m.loss_fn is torch.nn.functional.mse_loss)

    def step(self, batch, weight: float):
        """Forward + loss + backward of one microbatch by graph replay.  Returns (outputs, loss) like the eager pair
        ``model(batch)`` / ``model.loss(...)`` followed by ``model.backward_from_loss()``."""
        m = self.model
        lat, cond = batch[m.image_latents_key], batch[m.text_latents_key]
        key = (tuple(lat.shape), lat.dtype, tuple(cond.shape), cond.dtype, round(float(weight), 9), m.prediction_type)
        g = self.graphs.get(key)
        dev = m.unet.device_
        B = lat.shape[0]
        # the draws of stable_diffusion.py:177-179, from the same global generator as the eager path
        t = batch.get('_timesteps')
        if t is None:
            t = torch.randint(0, len(m.noise_scheduler), (B,), device=dev)
        noise = batch.get('_noise')
        if noise is None:
            noise = torch.randn(lat.shape, device=dev, dtype=lat.dtype if lat.is_floating_point() else torch.float32)
        if g is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            g = self.graphs[key] = GraphedMicrobatch(m, lat, cond, weight)
        return g.replay(lat, cond, t, noise)
