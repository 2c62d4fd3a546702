"""Reader (and a minimal writer, for tests) of MosaicML-streaming "MDS" shards - the on-disk format the reference's
precomputed latents live in (written by /root/reference scripts/precompute_latents.py:252-328 through
``streaming.MDSWriter``; read by diffusion/datasets/laion/laion.py:81-112 through ``streaming.StreamingDataset``).

``mosaicml-streaming`` is not installed here, so the format is restated from its published layout (format "mds",
version 2, no compression - the reference writes ``compression=None``, precompute_latents.py:275):
  directory/index.json : {"version": 2, "shards": [{"column_names", "column_encodings", "column_sizes" (null = variable),
                          "samples", "raw_data": {"basename"}, "compression": null, "format": "mds", ...}]}
  shard file           : uint32 num_samples | uint32 offsets[num_samples+1] (ABSOLUTE, from file start) |
                         config blob (the writer repeats the shard's column table as JSON here; the offsets skip it) |
                         sample blobs
  sample blob          : uint32 size for every variable-size column (in column order) | column payloads in order
  column order         : the writer sorts the columns by NAME; readers must follow index.json's ``column_names``
  encodings used here  : bytes, str (utf-8), int8..int64 / uint8..uint64 / float16..float64 (numpy scalars), int (int64)
No file produced by the real package is available offline; besides the round trip through ``write_mds`` the reader is
tested on a shard assembled byte by byte from this layout with ``struct`` (tests/test_abi_and_host.py), including the
config blob, name-sorted columns and the b'' latents of images below the resolution."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np

_SCALARS = {f'{k}{b}': np.dtype(f'{k}{b}') for k in ('int', 'uint') for b in (8, 16, 32, 64)}
_SCALARS.update({f'float{b}': np.dtype(f'float{b}') for b in (16, 32, 64)})
_SCALARS['int'] = np.dtype('int64')


def _fixed_size(enc: str) -> Optional[int]:
    return _SCALARS[enc].itemsize if enc in _SCALARS else None


def _decode(enc: str, raw: bytes):
    if enc == 'bytes' or enc in ('jpeg', 'png', 'pil'):
        return raw
    if enc == 'str':
        return raw.decode('utf-8')
    if enc in _SCALARS:
        return np.frombuffer(raw, _SCALARS[enc])[0]
    raise ValueError(f'unsupported MDS encoding {enc!r}')


def _encode(enc: str, value) -> bytes:
    if enc == 'bytes':
        return bytes(value)
    if enc == 'str':
        return str(value).encode('utf-8')
    if enc in _SCALARS:
        return np.asarray(value, _SCALARS[enc]).tobytes()
    raise ValueError(f'unsupported MDS encoding {enc!r}')


class MDSShard:

    def __init__(self, directory: str, info: dict):
        if info.get('compression'):
            raise ValueError('compressed MDS shards are not supported')
        self.path = os.path.join(directory, info['raw_data']['basename'])
        self.names: List[str] = info['column_names']
        self.encodings: List[str] = info['column_encodings']
        self.sizes: List[Optional[int]] = info['column_sizes']
        self.samples: int = info['samples']
        self._mm = None

    def _map(self):
        if self._mm is None:
            self._mm = np.memmap(self.path, dtype=np.uint8, mode='r')  # streamed, never loaded whole
            n = int(self._mm[:4].view(np.uint32)[0])
            if n != self.samples:
                raise ValueError(f'{self.path}: header says {n} samples, index says {self.samples}')
            self._offsets = self._mm[4:4 + 4 * (n + 1)].view(np.uint32)
        return self._mm

    def get(self, idx: int, columns=None) -> Dict[str, object]:
        mm = self._map()
        begin, end = int(self._offsets[idx]), int(self._offsets[idx + 1])
        data = bytes(mm[begin:end])
        sizes, pos = [], 0
        for size in self.sizes:
            if size:
                sizes.append(size)
            else:
                sizes.append(int(np.frombuffer(data[pos:pos + 4], np.uint32)[0]))
                pos += 4
        out = {}
        for name, enc, size in zip(self.names, self.encodings, sizes):
            if columns is None or name in columns:
                out[name] = _decode(enc, data[pos:pos + size])
            pos += size
        return out


class MDSDirectory:
    """All shards of one MDS directory; ``len`` / ``get(i)`` over the concatenated samples."""

    def __init__(self, directory: str):
        with open(os.path.join(directory, 'index.json')) as f:
            index = json.load(f)
        self.shards = [MDSShard(directory, s) for s in index['shards']]
        self._cum = np.cumsum([0] + [s.samples for s in self.shards])

    def __len__(self):
        return int(self._cum[-1])

    def get(self, idx: int, columns=None):
        si = int(np.searchsorted(self._cum, idx, side='right') - 1)
        return self.shards[si].get(idx - int(self._cum[si]), columns)


def write_mds(directory: str, columns: Dict[str, str], samples: List[dict], samples_per_shard: int = 1 << 30):
    """Minimal writer of the same layout (tests / local conversion of latents): columns sorted by name and the config
    blob in front of the samples, as ``streaming.MDSWriter`` lays a shard out; no compression, no hashes."""
    os.makedirs(directory, exist_ok=True)
    names = sorted(columns)
    encs = [columns[n] for n in names]
    sizes = [_fixed_size(e) for e in encs]
    config = json.dumps({'column_encodings': encs, 'column_names': names, 'column_sizes': sizes, 'compression': None,
                         'format': 'mds', 'hashes': [], 'size_limit': None, 'version': 2}, sort_keys=True).encode('utf-8')
    shards = []
    for si, start in enumerate(range(0, len(samples), samples_per_shard)):
        chunk = samples[start:start + samples_per_shard]
        blobs = []
        for smp in chunk:
            payload = [_encode(e, smp[n]) for n, e in zip(names, encs)]
            head = b''.join(np.uint32(len(p)).tobytes() for p, s in zip(payload, sizes) if s is None)
            blobs.append(head + b''.join(payload))
        n = len(chunk)
        header = 4 + 4 * (n + 1) + len(config)
        offs = np.cumsum([header] + [len(b) for b in blobs]).astype(np.uint32)
        base = f'shard.{si:05d}.mds'
        with open(os.path.join(directory, base), 'wb') as f:
            f.write(np.uint32(n).tobytes() + offs.tobytes() + config + b''.join(blobs))
        shards.append({'column_encodings': encs, 'column_names': names, 'column_sizes': sizes, 'compression': None,
                       'format': 'mds', 'hashes': [], 'raw_data': {'basename': base, 'bytes': int(offs[-1]), 'hashes': {}},
                       'samples': n, 'size_limit': None, 'version': 2, 'zip_data': None})
    with open(os.path.join(directory, 'index.json'), 'w') as f:
        json.dump({'shards': shards, 'version': 2}, f)
