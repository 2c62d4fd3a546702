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
from torch.utils.data import DataLoader, Dataset, DistributedSampler, SequentialSampler


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
        self.skipped = 0

    MAX_SKIP = 4096

    def __len__(self):
        return len(self.mds)

    def __getitem__(self, index):
        s = self.image_size // 8
        n = len(self.mds)
        # the writer stores b'' when the source image is smaller than the resolution (precompute_latents.py:303-306):
        # such a sample cannot be trained on at this resolution - take the next one that can (deterministic in index)
        for probe in range(self.MAX_SKIP):
            smp = self.mds.get((index + probe) % n, columns=('caption', 'caption_latents', self.col))
            lat = np.frombuffer(smp[self.col], dtype=np.float16)
            if lat.size == 4 * s * s:
                break
            self.skipped += 1
        else:
            raise RuntimeError(f'{self.MAX_SKIP} consecutive samples from index {index} have no {self.col}: '
                               f'this MDS directory holds no latents for {self.image_size} px')
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
    remote, local = _as_list(remote), _as_list(local)
    if remote and local and len(remote) != len(local):
        raise ValueError(f'remote and local Sequences must be the same length, got lengths {len(remote)} and {len(local)}')
    if remote and not local:
        raise ValueError('a remote without a local cache directory cannot be read here: mosaicml-streaming (download-on-'
                         'miss) is not available; point `local` at a directory that already holds the shards')
    text_dim = dataloader_kwargs.pop('text_dim', 1024)
    with_images = dataloader_kwargs.pop('synthetic_images', False)
    seed = int(dataloader_kwargs.pop('seed', 17))
    if local:
        missing = [d for d in local if not os.path.isdir(d)]
        if missing:  # never fall through to synthetic noise because of a mistyped path
            raise FileNotFoundError(f'local dataset director{"ies" if len(missing) > 1 else "y"} not found: {missing}')
        from ...models.text import build_tokenizer
        tok = build_tokenizer(tokenizer_name_or_path if os.path.isdir(str(tokenizer_name_or_path)) else None)
        parts = [MDSLatentDataset(d, resize_size, tok, caption_drop_prob) if os.path.exists(os.path.join(d, 'index.json'))
                 else LocalLatentShards(d, resize_size) for d in local]
        dataset = torch.utils.data.ConcatDataset(parts)
    else:  # both remote and local empty (the shipped YAML): seeded synthetic data of the same shapes
        dataset = SyntheticLAIONDataset(image_size=resize_size, caption_drop_prob=caption_drop_prob, text_dim=text_dim,
                                        with_images=with_images, seed=seed)
    if num_samples is not None:
        dataset = torch.utils.data.Subset(dataset, range(num_samples))
    if dataloader_kwargs.get('num_workers', 0) == 0:
        dataloader_kwargs.pop('prefetch_factor', None)
        dataloader_kwargs.pop('persistent_workers', None)
    # The reference gets shuffling and the per-rank partition from StreamingDataset (laion.py:167-180: shuffle=...,
    # batch_size=..., num_canonical_nodes=...) with train.py:40 dividing the batch by the world size.  Here: a
    # rank-strided partition of a per-epoch seeded permutation (torch DistributedSampler), so the ranks of a data-
    # parallel job read disjoint, jointly exhaustive samples; single process: a seeded RandomSampler.
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if world > 1:
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed,
                                     drop_last=drop_last)
    elif shuffle:
        sampler = EpochRandomSampler(dataset, seed=seed)
    else:
        sampler = SequentialSampler(dataset)
    return EpochDataLoader(dataset=dataset, batch_sampler=ResumableBatchSampler(sampler, batch_size, drop_last),
                           **dataloader_kwargs)


def _as_list(x) -> List[str]:
    """None / '' -> [], 'path' -> ['path'], sequences -> list without empty entries."""
    if x is None:
        return []
    if isinstance(x, (str, os.PathLike)):
        return [str(x)] if str(x) else []
    return [str(d) for d in x if d]


class EpochRandomSampler(torch.utils.data.Sampler):
    """Single-process shuffle: the permutation of epoch e is a pure function of (seed, e) - like DistributedSampler's -
    so a resumed run reproduces the epoch it stopped in instead of continuing a generator it no longer has."""

    def __init__(self, data_source, seed: int = 0):
        self.n = len(data_source)
        self.seed = int(seed)
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        return iter(torch.randperm(self.n, generator=g).tolist())

    def __len__(self):
        return self.n


class ResumableBatchSampler(torch.utils.data.Sampler):
    """torch's BatchSampler plus ``skip``: the next iterator drops its first ``skip`` batches at INDEX level (no sample is
    loaded for them) - how a resumed run continues in the middle of an epoch (streaming's StreamingDataset resumes
    mid-epoch from its state dict; reference laion.py:167-180 / SD-2-base-256.yaml:91-94 autoresume)."""

    def __init__(self, sampler, batch_size: int, drop_last: bool):
        self.sampler, self.batch_size, self.drop_last = sampler, int(batch_size), bool(drop_last)
        self.skip = 0

    def __iter__(self):
        skip, self.skip = self.skip, 0
        batch, nb = [], 0
        for idx in self.sampler:
            batch.append(idx)
            if len(batch) == self.batch_size:
                if nb >= skip:
                    yield batch
                nb += 1
                batch = []
        if batch and not self.drop_last and nb >= skip:
            yield batch

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)


class EpochDataLoader(DataLoader):
    """DataLoader that advances its sampler's epoch each time a new iterator is made (the samplers reshuffle only when
    told the epoch).  ``set_epoch(e, skip_batches=k)`` lets a resumed run continue the sequence: the next iterator is
    epoch e without its first k batches."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._next_epoch = 0

    @property
    def index_sampler(self):
        return self.batch_sampler.sampler if isinstance(self.batch_sampler, ResumableBatchSampler) else self.sampler

    def set_epoch(self, epoch: int, skip_batches: int = 0):
        self._next_epoch = int(epoch)
        if isinstance(self.batch_sampler, ResumableBatchSampler):
            self.batch_sampler.skip = int(skip_batches)
        elif skip_batches:
            raise ValueError('skip_batches needs a ResumableBatchSampler')

    def __iter__(self):
        smp = self.index_sampler
        if hasattr(smp, 'set_epoch'):
            smp.set_epoch(self._next_epoch)
        self._next_epoch += 1
        return super().__iter__()
