"""Constructors for diffusion models - mirrors /root/reference diffusion/models/models.py:28-112.

``stable_diffusion_2`` keeps the reference signature (:28-39).  Differences forced by the platform:
  * nothing is fetched: the SD-2-base / SD-2.1 U-Net configs are embedded (``unet.UNetConfig``); ``model_name`` may
    be a LOCAL directory with ``unet/``, ``vae/``, ``text_encoder/``, ``tokenizer/`` sub-folders holding
    safetensors / vocab files, otherwise ``pretrained=True`` raises;
  * the U-Net is ``UNetHIP`` (hand-written gfx950 kernels) instead of ``diffusers.UNet2DConditionModel``;
  * the frozen VAE / text encoder are stock PyTorch-ROCm modules (random-init without local weights) and are only
    built when they can be needed (``precomputed_latents=False``) or when asked for with ``build_encoders=True``;
  * xformers (:109-111) is not a concept here: attention is always the fused flash kernels."""
from __future__ import annotations

import os
import warnings
from typing import List, Optional

import torch

from .composer_shim import MeanSquaredError
from .schedulers import DDIMScheduler, DDPMScheduler
from .stable_diffusion import StableDiffusion
from .unet import UNetConfig, UNetHIP

_KNOWN = {
    'stabilityai/stable-diffusion-2-base': UNetConfig.sd2_base,
    'stabilityai/stable-diffusion-2-1-base': UNetConfig.sd2_base,
    'stabilityai/stable-diffusion-2': UNetConfig.sd21_768v,
    'stabilityai/stable-diffusion-2-1': UNetConfig.sd21_768v,
    'tiny': UNetConfig.tiny,
}


def _load_local_unet_weights(unet: UNetHIP, directory: str):
    from safetensors.torch import load_file
    for fn in ('diffusion_pytorch_model.safetensors', 'model.safetensors'):
        path = os.path.join(directory, 'unet', fn)
        if os.path.exists(path):
            unet.load_state_dict(load_file(path))
            return
    raise FileNotFoundError(f'no safetensors U-Net weights under {directory}/unet')


# diffusers < 0.17 checkpoints (the published SD-2 VAE files among them) name the mid-block attention projections
# query / key / value / proj_attn; later versions to_q / to_k / to_v / to_out.0.  Accept both.
_VAE_ATTN_RENAMES = (('.query.', '.to_q.'), ('.key.', '.to_k.'), ('.value.', '.to_v.'), ('.proj_attn.', '.to_out.0.'))


def load_local_vae_weights(vae, directory: str):
    """Strictly load ``<directory>/vae/*.safetensors`` into the PyTorch-ROCm AutoencoderKL (reference: models.py:80-85
    always loads pretrained VAE weights)."""
    from safetensors.torch import load_file
    for fn in ('diffusion_pytorch_model.safetensors', 'model.safetensors'):
        path = os.path.join(directory, 'vae', fn)
        if os.path.exists(path):
            sd = {}
            for k, v in load_file(path).items():
                if '.attentions.' in k:
                    for old, new in _VAE_ATTN_RENAMES:
                        k = k.replace(old, new)
                    if v.dim() == 4 and v.shape[-2:] == (1, 1) and ('.to_' in k):
                        v = v[:, :, 0, 0]  # very old checkpoints store the projections as 1x1 convolutions
                sd[k] = v
            vae.load_state_dict(sd, strict=True)
            return path
    raise FileNotFoundError(f'no safetensors VAE weights under {directory}/vae')


def stable_diffusion_2(
    model_name: str = 'stabilityai/stable-diffusion-2-base',
    pretrained: bool = True,
    train_metrics: Optional[List] = None,
    val_metrics: Optional[List] = None,
    val_guidance_scales: Optional[List] = None,
    val_seed: int = 1138,
    loss_bins: Optional[List] = None,
    precomputed_latents: bool = False,
    encode_latents_in_fp16: bool = True,
    fsdp: bool = True,
    build_encoders: Optional[bool] = None,
    unet_config: Optional[UNetConfig] = None,
    seed: int = 17,
):
    if train_metrics is None:
        train_metrics = [MeanSquaredError()]
    if val_metrics is None:
        val_metrics = [MeanSquaredError()]
    if val_guidance_scales is None:
        val_guidance_scales = [1.0, 3.0, 7.0]
    if loss_bins is None:
        loss_bins = [(0, 1)]
    if not torch.cuda.is_available():
        raise RuntimeError('stable_diffusion_2: an MI355X is required (the U-Net has no CPU path)')

    local = model_name if os.path.isdir(model_name) else None
    if unet_config is None:
        if model_name in _KNOWN:
            unet_config = _KNOWN[model_name]()
        elif local:
            unet_config = UNetConfig.sd2_base()
        else:
            raise ValueError(f'unknown model_name {model_name!r}: known names {sorted(_KNOWN)} or a local directory')
    unet = UNetHIP(unet_config, device='cuda', seed=seed, init=True)
    if pretrained:
        if not local:
            raise RuntimeError('pretrained=True needs model_name to be a local checkpoint directory '
                               '(no network / HF hub in this environment)')
        _load_local_unet_weights(unet, local)

    if build_encoders is None:
        build_encoders = not precomputed_latents
    vae = text_encoder = vae_hip = text_hip = None
    from .text import build_text_encoder, build_tokenizer
    tokenizer = build_tokenizer(os.path.join(local, 'tokenizer') if local else None)
    if build_encoders:
        from .vae import AutoencoderKL
        dtype = torch.float16 if encode_latents_in_fp16 else torch.float32
        vae = AutoencoderKL()
        te_dir = os.path.join(local, 'text_encoder') if local else None
        if local and os.path.isdir(os.path.join(local, 'vae')):
            load_local_vae_weights(vae, local)
        elif pretrained:
            raise FileNotFoundError(f'pretrained=True but {local}/vae holds no weights: the frozen VAE would be random '
                                    '(reference models.py:80-85 always loads it)')
        else:
            warnings.warn('AutoencoderKL is RANDOM-INIT (no local checkpoint directory): fine for throughput runs and '
                          'tests, meaningless for real training with precomputed_latents=False or for generate()')
        if pretrained and not (te_dir and os.path.isdir(te_dir)):
            raise FileNotFoundError(f'pretrained=True but {te_dir} is missing: the frozen text encoder would be random')
        vae_hip = None
        # The HIP encoders compute in bf16 (activations and residual stream), i.e. at the precision class the caller asked
        # for with encode_latents_in_fp16=True (the reference default, models.py:37).  With encode_latents_in_fp16=False
        # the reference encodes the training targets in fp32: then the PyTorch-ROCm fp32 modules run unless the HIP
        # encoders are asked for explicitly (DA_VAE_HIP=1 / DA_TEXT_HIP=1).
        hip_default = '1' if encode_latents_in_fp16 else '0'
        if os.environ.get('DA_VAE_HIP', hip_default) != '0':
            # the encoder half of the frozen VAE on the HIP kernels (models/vae_hip.py), built from the fp32 weights
            from .vae_hip import VAEEncoderHIP
            vae_hip = VAEEncoderHIP(vae.to('cuda'))
        vae = vae.to('cuda', dtype)
        text_encoder = build_text_encoder(te_dir, torch.float32, hidden_size=unet_config.cross_attention_dim).to('cuda')
        text_hip = None
        if os.environ.get('DA_TEXT_HIP', hip_default) != '0' and text_encoder.config.hidden_size % 64 == 0 and \
                text_encoder.config.hidden_size // text_encoder.config.num_attention_heads == 64:
            # the frozen text encoder on the HIP kernels (models/text_hip.py), built from the fp32 weights
            from .text_hip import TextEncoderHIP
            text_hip = TextEncoderHIP(text_encoder)
        text_encoder = text_encoder.to(dtype)
    noise_scheduler = DDPMScheduler(prediction_type=unet_config.prediction_type)
    inference_noise_scheduler = DDIMScheduler(prediction_type=unet_config.prediction_type)

    model = StableDiffusion(
        unet=unet,
        vae=vae,
        text_encoder=text_encoder,
        tokenizer=tokenizer,
        noise_scheduler=noise_scheduler,
        inference_noise_scheduler=inference_noise_scheduler,
        train_metrics=train_metrics,
        val_metrics=val_metrics,
        val_guidance_scales=val_guidance_scales,
        val_seed=val_seed,
        loss_bins=loss_bins,
        precomputed_latents=precomputed_latents,
        encode_latents_in_fp16=encode_latents_in_fp16,
        fsdp=fsdp,
    )
    model.vae_hip = vae_hip if build_encoders else None
    model.text_hip = text_hip if build_encoders else None
    return model
