"""Frozen OpenCLIP-H text encoder + tokenizer on stock PyTorch-ROCm / transformers (no hub access).

Stands where diffusion/models/models.py:82,85,87 load ``CLIPTextModel`` / ``CLIPTokenizer`` by hub NAME.  Here the
SD-2 text-encoder config is embedded (hidden 1024, 23 layers, 16 heads, 77 positions, vocab 49408 - the published
stabilityai/stable-diffusion-2-base text_encoder/config.json) and the model is random-init unless ``local_dir``
points at a local checkpoint directory.  Without vocabulary files the real BPE tokenizer cannot be built; the
``ByteTokenizer`` below keeps the call surface (``__call__``, ``model_max_length``, ``decode``) for synthetic runs."""
from __future__ import annotations

import os
from typing import List, Optional, Union

import torch


def build_text_encoder(local_dir: Optional[str] = None, dtype=torch.float32, num_hidden_layers: int = 23,
                       hidden_size: int = 1024):
    from transformers import CLIPTextConfig, CLIPTextModel
    if local_dir and os.path.isdir(local_dir):
        return CLIPTextModel.from_pretrained(local_dir, torch_dtype=dtype, local_files_only=True)
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=hidden_size, intermediate_size=4 * hidden_size,
                         num_hidden_layers=num_hidden_layers, num_attention_heads=max(1, hidden_size // 64),
                         max_position_embeddings=77, hidden_act='gelu', projection_dim=512, pad_token_id=1,
                         bos_token_id=0, eos_token_id=2)
    return CLIPTextModel(cfg).to(dtype)


class ByteTokenizer:
    model_max_length = 77
    bos, eos = 49406, 49407

    def __call__(self, text: Union[str, List[str]], padding='max_length', max_length: Optional[int] = None,
                 truncation=True, return_tensors: Optional[str] = None):
        max_length = max_length or self.model_max_length
        texts = [text] if isinstance(text, str) else list(text)
        ids = []
        for t in texts:
            body = [b + 256 for b in t.encode('utf-8')][:max_length - 2]
            row = [self.bos] + body + [self.eos]
            row += [self.eos] * (max_length - len(row))
            ids.append(row)
        if isinstance(text, str) and return_tensors is None:
            ids = ids[0]

        class _Enc(dict):
            pass

        out = _Enc(input_ids=torch.tensor(ids) if return_tensors == 'pt' else ids)
        out.input_ids = out['input_ids']
        return out

    def decode(self, ids, skip_special_tokens=True):
        bs = bytes(int(i) - 256 for i in ids if 256 <= int(i) < 512)
        return bs.decode('utf-8', errors='ignore')


def build_tokenizer(local_dir: Optional[str] = None):
    if local_dir and os.path.isdir(local_dir):
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(local_dir, local_files_only=True)
    return ByteTokenizer()
