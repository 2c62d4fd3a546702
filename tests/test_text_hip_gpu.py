"""The frozen text encoder on the HIP kernels (models/text_hip.py) against transformers.CLIPTextModel with the same weights:
last hidden state ``text_encoder(ids)[0]`` (reference stable_diffusion.py:168,172).  Tolerance: rel-L2 <= 2e-2 against the
fp32 torch module (bf16 activations through up to 23 pre-LN layers, fp32 accumulation and statistics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


@pytest.mark.parametrize('hidden,layers,B', [(1024, 23, 3), (128, 2, 5), (1024, 4, 64)])
def test_last_hidden_state_matches_torch(dev, hidden, layers, B):
    from diffusion_amd.models.text import build_text_encoder
    from diffusion_amd.models.text_hip import TextEncoderHIP
    torch.manual_seed(hidden + layers)
    te = build_text_encoder(None, torch.float32, num_hidden_layers=layers, hidden_size=hidden).to(dev).eval()
    with torch.no_grad():   # non-trivial norm affines and biases (torch default: gamma 1, beta 0, zero-ish biases)
        for n, p in te.named_parameters():
            if 'norm' in n or n.endswith('bias'):
                p.add_(0.1 * torch.randn_like(p))
    hip = TextEncoderHIP(te)
    ids = torch.randint(0, 49408, (B, 77), generator=torch.Generator().manual_seed(B)).to(dev)
    with torch.no_grad():
        ref = te(ids)[0]
    got = hip(ids)[0]
    assert got.shape == ref.shape == (B, 77, hidden) and got.dtype == torch.float32
    assert _rel(got, ref) < 2e-2, _rel(got, ref)
    # causal: the state of token t must not depend on later tokens
    ids2 = ids.clone(); ids2[:, 40:] = 7
    got2 = hip(ids2)[0]
    assert torch.equal(got[:, :40], got2[:, :40])


def test_factory_routes_text_through_hip(dev):
    from diffusion_amd.models.models import stable_diffusion_2
    m = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=False, encode_latents_in_fp16=True)
    if m.text_hip is None:
        pytest.skip('tiny config: text encoder head_dim != 64')
    ids = torch.randint(0, 1000, (2, 77)).to(dev)
    a = m._text_states(ids)
    with torch.no_grad():
        b = m.text_encoder(ids)[0]
    assert _rel(a, b) < 3e-2
