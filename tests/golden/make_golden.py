"""Generates the committed fixtures under tests/golden/ from the oracle itself (run in the build container):
  sd2_base_manifest.json  - the 686-entry diffusers-named (key, shape) manifest of the SD-2-base U-Net
  tiny_step_fp64.npz      - fp64 oracle training step on the tiny config (seed 17): inputs, eps-prediction, loss and
                            three parameter gradients.
The reference cannot be imported here (composer / diffusers absent) and holds no golden vectors of its own
(SURVEY.md 8c), so these pin the ORACLE against drift; they are not reference outputs."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import unet_oracle as O  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
man = O.param_manifest(O.UNetConfig.sd2_base())
with open(os.path.join(here, 'sd2_base_manifest.json'), 'w') as f:
    json.dump([[k, list(s)] for k, s in man], f)

cfg = O.UNetConfig.tiny()
sd = O.init_state_dict(cfg, seed=17, dtype=torch.float64)
g = torch.Generator().manual_seed(17)
B, S = 2, 8
lat = torch.randn(B, 4, S, S, generator=g, dtype=torch.float64)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g, dtype=torch.float64)
noise = torch.randn(B, 4, S, S, generator=g, dtype=torch.float64)
t = torch.randint(0, 1000, (B,), generator=g)
loss, pred, grads = O.training_loss_and_grads(sd, cfg, lat, t, ctx, noise)
out = dict(latents=lat.numpy(), ctx=ctx.numpy(), noise=noise.numpy(), t=t.numpy(), loss=loss.numpy(), pred=pred.numpy())
for k in ('conv_in.weight', 'mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight', 'conv_out.bias'):
    out['grad.' + k] = grads[k].numpy()
np.savez_compressed(os.path.join(here, 'tiny_step_fp64.npz'), **out)
print('wrote fixtures; loss =', float(loss))
