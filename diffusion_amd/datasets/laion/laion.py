"""LAION dataloader surface - mirrors /root/reference diffusion/datasets/laion/laion.py:115-194.

``build_streaming_laion_dataloader`` keeps the reference signature (:115-130) and the batch-dict contract of
``StreamingLAIONDataset.__getitem__`` (:81-112): ``image`` 3xRxR fp32 in [-1,1], ``captions`` 77 int64,
``caption_latents`` 77x1024 fp16, ``image_latents`` 4x(R/8)x(R/8) fp16.  mosaicml-streaming (MDS) is not available
here, so two backends exist:
  * ``local`` pointing at an MDS directory (``index.json`` + ``shard.*.mds``, read by ``datasets/mds.py`` - the
    reference's own precomputed-latent format) or at a directory of ``*.npz`` shards with raw-fp16 columns ``caption_latents``,
    ``latents_256`` / ``latents_512`` (the column names scripts/precompute_latents.py:252-272 writes) and
    optional ``captions``;
  * no ``remote``/``local`` (the YAML default: both empty) -> a seeded synthetic dataset of the same shapes
    (N(0,1) latents / text embeddings), which is what bench.py and the tests use.
JPEG decode (:83) is skipped when latents are present: the reference decodes images it never uses."""
from __future__ import annotations

import glob
import os
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticLAIONDataset(Dataset):

    def __init__(self, num_samples: int = 1 << 20, image_size: int = 256, caption_drop_prob: float = 0.0, seed: int = 17,
                 text_dim: int = 1024, with_images: bool = False):
        self.n, self.image_size, self.seed = num_samples, image_size, seed
        self.text_dim, self.with_images, self.caption_drop_prob = text_dim, with_images, caption_drop_prob

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + index)
        s = self.image_size // 8
        out = {
            'captions': torch.randint(0, 49408, (77,), generator=g),
            'caption_latents': torch.randn(77, self.text_dim, generator=g).half(),
            'image_latents': torch.randn(4, s, s, generator=g).half(),
        }
        if self.with_images:
            out['image'] = torch.rand(3, self.image_size, self.image_size, generator=g) * 2 - 1
        return out


class MDSLatentDataset(Dataset):
    """Precomputed-latent MDS shards (the reference's training data format).  Yields the reference's sample dict
    (laion.py:102-112) minus the decoded JPEG: with latents present the image is never used by ``forward``
    (stable_diffusion.py:157-158), and decoding it is the dominant host cost of the reference loader."""

    def __init__(self, directory: str, image_size: int, tokenizer=None, caption_drop_prob: float = 0.0):
        from ..mds import MDSDirectory
        self.mds = MDSDirectory(directory)
        self.image_size, self.tokenizer, self.caption_drop_prob = image_size, tokenizer, caption_drop_prob
        self.col = f'latents_{image_size}'

    def __len__(self):
        return len(self.mds)

    def __getitem__(self, index):
        smp = self.mds.get(index, columns=('caption', 'caption_latents', self.col))
        s = self.image_size // 8
        lat = np.frombuffer(smp[self.col], dtype=np.float16)
        if lat.size != 4 * s * s:
            # the writer stores b'' when the source image is smaller than the resolution (precompute_latents.py:303-306)
            raise IndexError(f'sample {index} has no {self.col}')
        out = {'caption_latents': torch.from_numpy(np.frombuffer(smp['caption_latents'], dtype=np.float16).copy()).reshape(77, -1),
               'image_latents': torch.from_numpy(lat.copy()).reshape(4, s, s)}
        caption = '' if torch.rand(1) < self.caption_drop_prob else smp.get('caption', '')
        if self.tokenizer is not None:
            ids = self.tokenizer(caption, padding='max_length', max_length=self.tokenizer.model_max_length,
                                 truncation=True)['input_ids']
            out['captions'] = torch.tensor(ids)
        else:
            out['captions'] = torch.zeros(77, dtype=torch.int64)
        return out


class LocalLatentShards(Dataset):
    """*.npz shards: arrays ``caption_latents`` [n,77*1024] fp16 bytes-equivalent, ``latents_{res}`` [n,4*s*s]."""

    def __init__(self, directory: str, image_size: int):
        self.files = sorted(glob.glob(os.path.join(directory, '*.npz')))
        if not self.files:
            raise FileNotFoundError(f'no *.npz latent shards under {directory}')
        self.image_size = image_size
        self.index = []
        for fi, f in enumerate(self.files):
            with np.load(f) as z:
                n = z['caption_latents'].shape[0]
            self.index += [(fi, i) for i in range(n)]
        self._cache = (None, None)

    def __len__(self):
        return len(self.index)

    def __getitem__(self, index):
        fi, i = self.index[index]
        if self._cache[0] != fi:
            self._cache = (fi, dict(np.load(self.files[fi])))
        z = self._cache[1]
        s = self.image_size // 8
        out = {
            'caption_latents': torch.from_numpy(z['caption_latents'][i].astype(np.float16).copy()).reshape(77, -1),
            'image_latents': torch.from_numpy(z[f'latents_{self.image_size}'][i].astype(np.float16).copy()).reshape(4, s, s),
        }
        out['captions'] = torch.from_numpy(z['captions'][i].astype(np.int64)) if 'captions' in z else torch.zeros(
            77, dtype=torch.int64)
        return out


def build_streaming_laion_dataloader(
    remote: Union[str, List, None] = None,
    local: Union[str, List, None] = None,
    batch_size: int = 1,
    tokenizer_name_or_path: str = 'stabilityai/stable-diffusion-2-base',
    caption_drop_prob: float = 0.0,
    resize_size: int = 256,
    num_samples: Optional[int] = None,
    predownload: int = 100_000,
    download_retry: int = 2,
    download_timeout: float = 120,
    drop_last: bool = True,
    shuffle: bool = True,
    num_canonical_nodes: Optional[int] = None,
    **dataloader_kwargs,
):
    if isinstance(remote, str) and isinstance(local, str):
        remote, local = [remote], [local]
    elif isinstance(remote, Sequence) and isinstance(local, Sequence) and len(remote) != len(local):
        raise ValueError(f'remote and local Sequences must be the same length, got lengths {len(remote)} and {len(local)}')
    text_dim = dataloader_kwargs.pop('text_dim', 1024)
    with_images = dataloader_kwargs.pop('synthetic_images', False)
    dirs = [d for d in (local or []) if d and os.path.isdir(d)]
    if dirs:
        from ...models.text import build_tokenizer
        tok = build_tokenizer(tokenizer_name_or_path if os.path.isdir(str(tokenizer_name_or_path)) else None)
        parts = [MDSLatentDataset(d, resize_size, tok, caption_drop_prob) if os.path.exists(os.path.join(d, 'index.json'))
                 else LocalLatentShards(d, resize_size) for d in dirs]
        dataset = torch.utils.data.ConcatDataset(parts)
    else:
        dataset = SyntheticLAIONDataset(image_size=resize_size, caption_drop_prob=caption_drop_prob, text_dim=text_dim,
                                        with_images=with_images)
    if num_samples is not None:
        dataset = torch.utils.data.Subset(dataset, range(num_samples))
    if dataloader_kwargs.get('num_workers', 0) == 0:
        dataloader_kwargs.pop('prefetch_factor', None)
        dataloader_kwargs.pop('persistent_workers', None)
    return DataLoader(dataset=dataset, batch_size=batch_size, sampler=None, drop_last=drop_last, **dataloader_kwargs)
