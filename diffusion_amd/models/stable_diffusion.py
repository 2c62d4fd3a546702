"""``StableDiffusion`` ComposerModel on the HIP U-Net.

Mirrors /root/reference diffusion/models/stable_diffusion.py: constructor kwargs (:69-88), ``forward`` (:154-183),
``loss`` (:185-187), ``eval_forward`` (:189-208), ``get_metrics`` (:210-226), ``update_metric`` (:228-257),
``generate`` (:260-382).  Differences, all on the device side:
  * ``add_noise`` (:180), the NCHW->NHWC relayout, the U-Net (:183), ``F.mse_loss`` (:187) and the whole backward
    run as hand-written gfx950 kernels; RNG draws (:177,179) stay on torch's global generator like the reference.
  * ``loss()`` returns a 0-d tensor whose ``.backward()`` runs the HIP backward into the U-Net's flat fp32
    gradient buffer (so Composer-style trainers work unchanged); ``backward_from_loss()`` is the direct call.
  * ``prediction_type='v_prediction'`` (docstring :40-43, never wired in the reference's forward) follows
    diffusion/models/pixel_diffusion.py:90-91.
"""
from __future__ import annotations

import copy
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import ops
from .composer_shim import ComposerModel, MeanSquaredError, Metric
from .unet import UNetHIP

try:
    from tqdm.auto import tqdm
except Exception:  # noqa: BLE001
    tqdm = lambda x, **kw: x  # noqa: E731


class _HIPBackward(torch.autograd.Function):
    """Gives the fused-loss scalar an autograd edge: ``loss.backward()`` -> UNetHIP.backward_features."""

    @staticmethod
    def forward(ctx, anchor, loss_value, model):
        ctx.model = model
        return loss_value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model._run_backward(g)
        return torch.zeros((), device=g.device), None, None


class StableDiffusion(ComposerModel):

    def __init__(self,
                 unet,
                 vae,
                 text_encoder,
                 tokenizer,
                 noise_scheduler,
                 inference_noise_scheduler,
                 loss_fn=F.mse_loss,
                 train_metrics: Optional[List] = None,
                 val_metrics: Optional[List] = None,
                 val_seed: int = 1138,
                 val_guidance_scales: Optional[List] = None,
                 loss_bins: Optional[List] = None,
                 image_key: str = 'image',
                 text_key: str = 'captions',
                 image_latents_key: str = 'image_latents',
                 text_latents_key: str = 'caption_latents',
                 precomputed_latents: bool = False,
                 encode_latents_in_fp16: bool = False,
                 fsdp: bool = False,
                 prediction_type: Optional[str] = None):
        super().__init__()
        self.unet = unet
        self.vae = vae
        self.noise_scheduler = noise_scheduler
        self.loss_fn = loss_fn
        self.val_seed = val_seed
        self.image_key = image_key
        self.image_latents_key = image_latents_key
        self.precomputed_latents = precomputed_latents
        self.prediction_type = prediction_type or getattr(noise_scheduler, 'prediction_type', 'epsilon')
        if self.prediction_type not in ('epsilon', 'v_prediction'):
            raise ValueError(f'prediction type must be epsilon or v_prediction. Got {self.prediction_type}')

        self.train_metrics = [MeanSquaredError()] if train_metrics is None else train_metrics
        if val_metrics is None:
            val_metrics = [MeanSquaredError()]
        if val_guidance_scales is None:
            val_guidance_scales = [0.0]
        if loss_bins is None:
            loss_bins = [(0, 1)]
        self.val_guidance_scales = val_guidance_scales
        self.val_metrics = {}
        metrics_to_sweep = ['FrechetInceptionDistance', 'InceptionScore', 'CLIPScore']
        for metric in val_metrics:
            name = metric.__class__.__name__
            if name in metrics_to_sweep:
                for scale in val_guidance_scales:
                    new_metric = copy.deepcopy(metric)
                    new_metric.guidance_scale = scale
                    self.val_metrics[f'{name}-scale-{str(scale).replace(".", "p")}'] = new_metric
            elif isinstance(metric, MeanSquaredError):
                for bin in loss_bins:
                    new_metric = copy.deepcopy(metric)
                    new_metric.loss_bin = bin
                    self.val_metrics[f'{name}-bin-{bin[0]}-to-{bin[1]}'.replace('.', 'p')] = new_metric
            else:
                self.val_metrics[name] = metric
        self.val_metrics['MeanSquaredError'] = MeanSquaredError()

        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.inference_scheduler = inference_noise_scheduler
        self.text_key = text_key
        self.text_latents_key = text_latents_key
        self.encode_latents_in_fp16 = encode_latents_in_fp16
        # freeze the encoders (reference :143-147)
        if self.text_encoder is not None:
            self.text_encoder.requires_grad_(False)
        if self.vae is not None:
            self.vae.requires_grad_(False)
        if self.encode_latents_in_fp16:
            if self.text_encoder is not None:
                self.text_encoder.half()
            if self.vae is not None:
                self.vae.half()
        if fsdp:
            if self.text_encoder is not None:
                self.text_encoder._fsdp_wrap = False
            if self.vae is not None:
                self.vae._fsdp_wrap = False
            self.unet._fsdp_wrap = True
        self._pending = None

    # ------------------------------------------------------------------------------------------
    def _text_states(self, input_ids):
        """``text_encoder(ids)[0]`` (reference :168,172): the HIP-kernel walk of the frozen weights when the factory built
        one (models/text_hip.py), else the PyTorch-ROCm module."""
        enc = getattr(self, 'text_hip', None)
        return (enc if enc is not None else self.text_encoder)(input_ids)[0]

    def _encode(self, batch):
        """latents / conditioning selection, reference :155-174."""
        if self.precomputed_latents and self.image_latents_key in batch and self.text_latents_key in batch:
            return batch[self.image_latents_key], batch[self.text_latents_key]
        if self.vae is None or self.text_encoder is None:
            raise RuntimeError('this model was built without VAE / text encoder; pass precomputed latents')
        inputs, conditioning = batch[self.image_key], batch[self.text_key]
        conditioning = conditioning.view(-1, conditioning.shape[-1])
        # image encoder: the HIP-kernel walk of the same frozen weights when the factory built one (models/vae_hip.py),
        # else the PyTorch-ROCm module
        vae_hip = getattr(self, 'vae_hip', None)
        with torch.no_grad():
            if vae_hip is not None:
                latents = vae_hip.encode(inputs)['latent_dist'].sample().data
                with torch.autocast('cuda', enabled=False):
                    conditioning = self._text_states(conditioning)
            elif self.encode_latents_in_fp16:
                with torch.autocast('cuda', enabled=False):
                    latents = self.vae.encode(inputs.half())['latent_dist'].sample().data
                    conditioning = self._text_states(conditioning)
            else:
                latents = self.vae.encode(inputs)['latent_dist'].sample().data
                conditioning = self._text_states(conditioning)
        latents *= 0.18215
        return latents, conditioning

    def forward(self, batch, timesteps: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """Returns ``(unet_out, target, timesteps)`` like the reference (:183).  ``timesteps`` / ``noise`` may be
        injected for parity tests (as arguments, or as ``batch['_timesteps']`` / ``batch['_noise']`` so that they pass
        through a trainer's microbatch slicing); by default they are drawn from torch's global RNG exactly as :177-179."""
        if timesteps is None:
            timesteps = batch.get('_timesteps')
        if noise is None:
            noise = batch.get('_noise')
        latents, conditioning = self._encode(batch)
        unet: UNetHIP = self.unet
        dev = unet.device_
        latents = latents.to(dev)
        B, _, S, S2 = latents.shape
        if timesteps is None:
            timesteps = torch.randint(0, len(self.noise_scheduler), (B,), device=dev)
        if noise is None:
            noise = torch.randn_like(latents)
        x0 = latents.float().contiguous()
        eps = noise.to(dev).float().contiguous()
        t = timesteps.to(dev, torch.int64).contiguous()
        sa, sb = self.noise_scheduler.device_tables(dev)
        xt = torch.empty(B * S * S2, 8, device=dev, dtype=torch.bfloat16)
        target8 = torch.empty(B * S * S2, 8, device=dev, dtype=torch.float32)
        v_pred = self.prediction_type == 'v_prediction'
        ops.add_noise(x0, eps, t, sa, sb, xt, target8, v_pred)
        ctx = unet.prepare_ctx(conditioning.to(dev))
        pred8 = unet.forward_features(xt, t, ctx, B, S)
        pred = pred8.view(B, S, S2, 8)[..., :4].permute(0, 3, 1, 2)
        target = target8.view(B, S, S2, 8)[..., :4].permute(0, 3, 1, 2) if v_pred else noise
        self._pending = (pred8, target8, B * S * S2)
        return pred, target, timesteps

    def loss(self, outputs, batch, weight: float = 1.0):
        """MSE between U-Net output and target (reference :185-187) - fused loss + gradient kernel.
        ``weight`` pre-scales the gradient (microbatch fraction) when the trainer calls backward directly."""
        if self._pending is None:
            raise RuntimeError('loss() must follow forward()')
        pred8, target8, npix = self._pending
        unet: UNetHIP = self.unet
        if self.loss_fn is F.mse_loss:
            dpred = torch.empty(npix, 8, device=pred8.device, dtype=torch.bfloat16)
            lossbuf = torch.zeros(1, device=pred8.device, dtype=torch.float32)
            ops.mse_loss(pred8, target8, dpred, lossbuf, unet._scratch, npix, 2.0 * weight / (4.0 * npix), 1.0, 0)
            self._dpred = dpred
            anchor = torch.zeros((), device=pred8.device, requires_grad=True)
            return _HIPBackward.apply(anchor, lossbuf[0], self)
        # user-supplied loss: scalar maths on torch autograd, U-Net backward still on the HIP path
        p = outputs[0].detach().float().requires_grad_(True)
        val = self.loss_fn(p, outputs[1].detach().float())
        (gp,) = torch.autograd.grad(val, p)
        self._dpred = unet.to_nhwc8(gp * weight)
        anchor = torch.zeros((), device=pred8.device, requires_grad=True)
        return _HIPBackward.apply(anchor, val.detach(), self)

    def _run_backward(self, g: Optional[torch.Tensor] = None):
        dpred, self._dpred = self._dpred, None
        if dpred is None:
            raise RuntimeError('backward already consumed')
        if g is not None:
            dpred = (dpred.float() * g).to(torch.bfloat16)  # [M,8] boundary scaling for external trainers
        self.unet.backward_features(dpred)
        self._pending = None

    def backward_from_loss(self):
        """Direct (no autograd) backward for the in-tree trainer; gradient weight was given to ``loss()``."""
        self._run_backward(None)

    # ------------------------------------------------------------------------------------------
    def eval_forward(self, batch, outputs=None):
        if outputs is not None:
            return outputs
        with torch.no_grad():
            unet_out, target, timesteps = self.forward(batch)
        self._pending = None
        self.unet._tape = None
        generated_images = {}
        if self.text_key in batch and self.image_key in batch and self.vae is not None:
            prompts = batch[self.text_key]
            height, width = batch[self.image_key].shape[-2], batch[self.image_key].shape[-1]
            for guidance_scale in self.val_guidance_scales:
                generated_images[guidance_scale] = self.generate(tokenized_prompts=prompts, height=height, width=width,
                                                                 guidance_scale=guidance_scale, seed=self.val_seed,
                                                                 progress_bar=False)
        return unet_out, target, timesteps, generated_images

    def get_metrics(self, is_train: bool = False):
        metrics = self.train_metrics if is_train else self.val_metrics
        if isinstance(metrics, Metric):
            return {metrics.__class__.__name__: metrics}
        if isinstance(metrics, list):
            # the reference keys every list entry by the list's class name (:219); keyed per metric here
            return {m.__class__.__name__: m for m in metrics}
        out = {}
        for name, metric in metrics.items():
            assert isinstance(metric, Metric)
            out[name] = metric
        return out

    def update_metric(self, batch, outputs, metric):
        if isinstance(metric, MeanSquaredError) and hasattr(metric, 'loss_bin'):
            loss_bin = metric.loss_bin
            T_max = self.noise_scheduler.num_train_timesteps
            idx = torch.where((outputs[2] >= loss_bin[0] * T_max) & (outputs[2] < loss_bin[1] * T_max))
            metric.update(outputs[0][idx], outputs[1][idx])
        elif isinstance(metric, MeanSquaredError):
            metric.update(outputs[0], outputs[1])
        elif metric.__class__.__name__ == 'FrechetInceptionDistance':
            metric.update(batch[self.image_key], real=True)
            metric.update(outputs[3][metric.guidance_scale], real=False)
        elif metric.__class__.__name__ == 'InceptionScore':
            metric.update(outputs[3][metric.guidance_scale])
        elif metric.__class__.__name__ == 'CLIPScore':
            captions = [self.tokenizer.decode(c, skip_special_tokens=True) for c in batch[self.text_key]]
            metric.update((outputs[3][metric.guidance_scale] * 255).to(torch.uint8), captions)
        else:
            metric.update(outputs[0], outputs[1])

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, prompt: Optional[list] = None, negative_prompt: Optional[list] = None,
                 tokenized_prompts: Optional[torch.LongTensor] = None,
                 tokenized_negative_prompts: Optional[torch.LongTensor] = None,
                 prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 3.0,
                 num_images_per_prompt: int = 1, seed: Optional[int] = None, progress_bar: bool = True):
        """DDIM sampling with classifier-free guidance on the HIP U-Net forward (reference :260-382)."""
        _check_prompt_given(prompt, tokenized_prompts, prompt_embeds)
        _check_prompt_lenths(prompt, negative_prompt)
        _check_prompt_lenths(tokenized_prompts, tokenized_negative_prompts)
        _check_prompt_lenths(prompt_embeds, negative_prompt_embeds)
        device = self.unet.device_
        rng = torch.Generator(device=device)
        if seed:
            rng = rng.manual_seed(seed)
        vae_scale = 8
        height = height or self.unet.config.sample_size * vae_scale
        width = width or self.unet.config.sample_size * vae_scale
        do_cfg = guidance_scale > 1.0
        text_embeddings = self._prepare_text_embeddings(prompt, tokenized_prompts, prompt_embeds, num_images_per_prompt)
        batch_size = len(text_embeddings)
        if do_cfg:
            if not negative_prompt and not tokenized_negative_prompts and not negative_prompt_embeds:
                negative_prompt = [''] * (batch_size // num_images_per_prompt)
            uncond = self._prepare_text_embeddings(negative_prompt, tokenized_negative_prompts, negative_prompt_embeds,
                                                   num_images_per_prompt)
            text_embeddings = torch.cat([uncond, text_embeddings])
        latents = torch.randn((batch_size, self.unet.config.in_channels, height // vae_scale, width // vae_scale),
                              device=device, generator=rng)
        self.inference_scheduler.set_timesteps(num_inference_steps)
        latents = latents * self.inference_scheduler.init_noise_sigma
        for t in tqdm(self.inference_scheduler.timesteps, disable=not progress_bar):
            lin = torch.cat([latents] * 2) if do_cfg else latents
            lin = self.inference_scheduler.scale_model_input(lin, t)
            pred = self.unet(lin, t, encoder_hidden_states=text_embeddings).sample
            if do_cfg:
                pu, pt = pred.chunk(2)
                pred = pu + guidance_scale * (pt - pu)
            latents = self.inference_scheduler.step(pred, t, latents, generator=rng)['prev_sample']
        latents = 1 / 0.18215 * latents
        vdtype = next(self.vae.parameters()).dtype
        image = self.vae.decode(latents.to(vdtype)).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.detach().float()

    def _prepare_text_embeddings(self, prompt, tokenized_prompts, prompt_embeds, num_images_per_prompt):
        device = self.unet.device_
        if prompt_embeds is None:
            if tokenized_prompts is None:
                tokenized_prompts = self.tokenizer(prompt, padding='max_length',
                                                   max_length=self.tokenizer.model_max_length, truncation=True,
                                                   return_tensors='pt').input_ids
            prompt_embeds = self._text_states(tokenized_prompts.to(device))
        prompt_embeds = prompt_embeds.to(device).float()
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1)
        return prompt_embeds.view(bs_embed * num_images_per_prompt, seq_len, -1)


def _check_prompt_lenths(prompt, negative_prompt):
    if prompt is None and negative_prompt is None:
        return
    batch_size = 1 if isinstance(prompt, str) else len(prompt)
    if negative_prompt:
        negative_prompt_bs = 1 if isinstance(negative_prompt, str) else len(negative_prompt)
        if negative_prompt_bs != batch_size:
            raise ValueError(f'len(prompts) and len(negative_prompts) must be the same. '
                             f'A negative prompt must be provided for each given prompt.')


def _check_prompt_given(prompt, tokenized_prompts, prompt_embeds):
    if prompt is None and tokenized_prompts is None and prompt_embeds is None:
        raise ValueError('Must provide one of `prompt`, `tokenized_prompts`, or `prompt_embeds`')
