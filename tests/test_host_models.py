"""CPU tests of host-side model plumbing that does not need the GPU: the frozen VAE's diffusers-compatible state_dict
(reference diffusion/models/models.py:80-85 loads ``AutoencoderKL.from_pretrained``) and local-weight loading."""
import os

import pytest
import torch


def _diffusers_vae_manifest():
    """Key -> shape of diffusers ``AutoencoderKL`` for the SD-2 VAE config (block_out_channels 128/256/512/512,
    layers_per_block 2, latent_channels 4), written out independently of diffusion_amd/models/vae.py."""
    m = {}

    def conv(k, co, ci, ks):
        m[k + '.weight'] = (co, ci, ks, ks)
        m[k + '.bias'] = (co,)

    def norm(k, c):
        m[k + '.weight'] = (c,)
        m[k + '.bias'] = (c,)

    def lin(k, co, ci):
        m[k + '.weight'] = (co, ci)
        m[k + '.bias'] = (co,)

    def res(k, ci, co):
        norm(k + '.norm1', ci); conv(k + '.conv1', co, ci, 3); norm(k + '.norm2', co); conv(k + '.conv2', co, co, 3)
        if ci != co:
            conv(k + '.conv_shortcut', co, ci, 1)

    def mid(k, c):
        res(k + '.resnets.0', c, c); res(k + '.resnets.1', c, c)
        a = k + '.attentions.0'
        norm(a + '.group_norm', c)
        for n in ('to_q', 'to_k', 'to_v', 'to_out.0'):
            lin(f'{a}.{n}', c, c)

    ch = (128, 256, 512, 512)
    conv('encoder.conv_in', 128, 3, 3)
    for i in range(4):
        ci = ch[max(i - 1, 0)]
        res(f'encoder.down_blocks.{i}.resnets.0', ci, ch[i]); res(f'encoder.down_blocks.{i}.resnets.1', ch[i], ch[i])
        if i < 3:
            conv(f'encoder.down_blocks.{i}.downsamplers.0.conv', ch[i], ch[i], 3)
    mid('encoder.mid_block', 512); norm('encoder.conv_norm_out', 512); conv('encoder.conv_out', 8, 512, 3)
    rev = (512, 512, 256, 128)
    conv('decoder.conv_in', 512, 4, 3); mid('decoder.mid_block', 512)
    for i in range(4):
        ci = rev[max(i - 1, 0)]
        for j in range(3):
            res(f'decoder.up_blocks.{i}.resnets.{j}', ci if j == 0 else rev[i], rev[i])
        if i < 3:
            conv(f'decoder.up_blocks.{i}.upsamplers.0.conv', rev[i], rev[i], 3)
    norm('decoder.conv_norm_out', 128); conv('decoder.conv_out', 3, 128, 3)
    conv('quant_conv', 8, 8, 1); conv('post_quant_conv', 4, 4, 1)
    return m


def test_vae_state_dict_matches_diffusers_manifest():
    from diffusion_amd.models.vae import AutoencoderKL
    vae = AutoencoderKL()
    sd = vae.state_dict()
    man = _diffusers_vae_manifest()
    assert set(sd) == set(man), sorted(set(sd) ^ set(man))[:10]
    for k, shape in man.items():
        assert tuple(sd[k].shape) == shape, k
    assert len(man) == 248 and sum(v.numel() for v in sd.values()) == 83_653_863   # public SD VAE size


def test_local_vae_weights_load_strictly(tmp_path):
    """A checkpoint in the published (pre-0.17 diffusers) naming - query/key/value/proj_attn - loads; a missing file or a
    missing tensor is an error (a pretrained run must never encode through a random VAE)."""
    from safetensors.torch import save_file
    from diffusion_amd.models.models import load_local_vae_weights
    from diffusion_amd.models.vae import AutoencoderKL
    torch.manual_seed(0)
    src = AutoencoderKL()
    old_names = {}
    for k, v in src.state_dict().items():
        for new, old in (('.to_q.', '.query.'), ('.to_k.', '.key.'), ('.to_v.', '.value.'), ('.to_out.0.', '.proj_attn.')):
            if '.attentions.' in k:
                k = k.replace(new, old)
        old_names[k] = v.contiguous()
    assert any('.proj_attn.' in k for k in old_names)
    d = tmp_path / 'ckpt'
    (d / 'vae').mkdir(parents=True)
    save_file(old_names, str(d / 'vae' / 'diffusion_pytorch_model.safetensors'))
    torch.manual_seed(1)
    dst = AutoencoderKL()
    load_local_vae_weights(dst, str(d))
    for (k, a), b in zip(src.state_dict().items(), dst.state_dict().values()):
        assert torch.equal(a, b), k
    x = torch.randn(1, 3, 32, 32)
    assert torch.equal(src.encode(x).latent_dist.mode(), dst.encode(x).latent_dist.mode())
    with pytest.raises(FileNotFoundError):
        load_local_vae_weights(dst, str(tmp_path / 'nothing-here'))
    del old_names['encoder.conv_in.bias']
    save_file(old_names, str(d / 'vae' / 'diffusion_pytorch_model.safetensors'))
    with pytest.raises(RuntimeError):
        load_local_vae_weights(dst, str(d))
