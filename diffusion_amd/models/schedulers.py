"""Noise schedulers with the diffusers surface the reference uses.

``DDPMScheduler`` stands where diffusion/models/models.py:88 loads ``diffusers.DDPMScheduler`` (hyper-parameters
restated in-tree at models.py:134-145); the reference touches ``len(scheduler)`` (stable_diffusion.py:177),
``add_noise`` (:180) and ``num_train_timesteps`` (:235).  ``get_velocity`` follows the use at
diffusion/models/pixel_diffusion.py:90-91.  ``DDIMScheduler`` (models.py:89) carries what ``generate()`` needs
(stable_diffusion.py:354-375): ``set_timesteps``, ``timesteps``, ``init_noise_sigma``, ``scale_model_input``, ``step``.
"""
from __future__ import annotations

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = 'scaled_linear', prediction_type: str = 'epsilon', clip_sample: bool = False):
        if beta_schedule != 'scaled_linear':
            raise ValueError('SD-2 uses the scaled_linear schedule')
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32)**2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self._dev_tables = {}

    def __len__(self):
        return self.num_train_timesteps

    def device_tables(self, device):
        """(sqrt(abar), sqrt(1-abar)) fp32 tables resident on `device` for the fused add_noise kernel."""
        key = str(device)
        if key not in self._dev_tables:
            ac = self.alphas_cumprod
            self._dev_tables[key] = ((ac**0.5).to(device).contiguous(), ((1 - ac)**0.5).to(device).contiguous())
        return self._dev_tables[key]

    # host-side reference forms (used by tests / non-HIP callers; the training path uses ops.add_noise)
    def _coef(self, t, like):
        ac = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)
        shape = (-1,) + (1,) * (like.dim() - 1)
        return (ac[t]**0.5).reshape(shape), ((1 - ac[t])**0.5).reshape(shape)

    def add_noise(self, original_samples, noise, timesteps):
        a, s = self._coef(timesteps, original_samples)
        return a * original_samples + s * noise

    def get_velocity(self, sample, noise, timesteps):
        a, s = self._coef(timesteps, sample)
        return a * noise - s * sample


class DDIMScheduler(DDPMScheduler):
    """eta = 0 DDIM sampler, ``set_alpha_to_one=False``, ``steps_offset=1`` (SD-2 scheduler_config.json)."""

    def __init__(self, *a, steps_offset: int = 1, **kw):
        super().__init__(*a, **kw)
        self.steps_offset = steps_offset
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.timesteps = torch.arange(self.num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(0, num_inference_steps) * ratio).flip(0) + self.steps_offset
        self.timesteps = ts.clamp(max=self.num_train_timesteps - 1).to(device) if device is not None else ts.clamp(
            max=self.num_train_timesteps - 1)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, generator=None, **kw):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        ac_t = self.alphas_cumprod[t].to(sample.device, sample.dtype)
        ac_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod).to(sample.device, sample.dtype)
        if self.prediction_type == 'v_prediction':
            x0 = ac_t.sqrt() * sample - (1 - ac_t).sqrt() * model_output
            eps = ac_t.sqrt() * model_output + (1 - ac_t).sqrt() * sample
        else:
            eps = model_output
            x0 = (sample - (1 - ac_t).sqrt() * eps) / ac_t.sqrt()
        prev = ac_prev.sqrt() * x0 + (1 - ac_prev).sqrt() * eps

        class _Out(dict):
            prev_sample = prev

        return _Out(prev_sample=prev)
