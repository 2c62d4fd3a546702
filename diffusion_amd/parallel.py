"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce of the flat fp32 gradient buffer
over RCCL/xGMI (``backend='nccl'`` is RCCL on ROCm), overlapped with the tail of backward.

The reference gets this from Composer + torch FSDP ``SHARD_GRAD_OP`` (yamls/hydra-yamls/SD-2-base-256.yaml:95-96,
batch split at diffusion/train.py:40).  On MI355X the 866 M-parameter replica (12 GB with Adam state) fits one GPU
many times over, so parameters and optimizer state are replicated and only gradients cross xGMI.

Why buckets of this shape: xGMI is a point-to-point mesh (7 links/GPU), collectives are per-link bound and have a
fixed launch/latency cost, so FEW LARGE buckets win; the flat buffer is laid out in forward order, backward
completes it back-to-front, and each bucket [lo, hi) is handed to RCCL on a side stream as soon as every gradient
at offsets >= lo is final.  The sum is left un-normalised: 1/world (and 1/microbatches) is folded into the fused
AdamW kernel's ``grad_scale``.

The exchange itself is selectable (constructor arguments or the environment):
  collective  'allreduce' (default) one ``all_reduce`` per bucket (RCCL picks ring / tree / direct for the mesh);
              'rs_ag'     ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` per bucket: on the xGMI full mesh
                          every rank sends shard j straight to peer j, all 7 links busy at once; with
                          the fp32 payload over RCCL both run IN PLACE on the flat gradient (no staging, no allocation)
  payload     'fp32' (default) the gradient as it lies; 'bf16' a bf16 staging copy (half the bytes on the links; the
              SUM is then rounded to 8 significant bits per element - opt-in).
  overlap     True (default) buckets are exchanged on the side stream while backward still runs; False: all buckets
              are exchanged after backward (``flush``).  Why one might want that: the GEMM kernels are launched as one
              workgroup per CU (persistent tile walks of exactly #CUs workgroups, weight-gradient grids sized to one
              round of the chip), and a workgroup of theirs needs a whole CU; every CU an RCCL channel occupies while
              such a launch starts pushes workgroups into a second round, so kernels that overlap a collective can
              take up to twice as long.  Unmeasured - no multi-GPU node was available to this build; the switch is
              here so that the first scaling run can decide.
  DA_DP_COLLECTIVE / DA_DP_PAYLOAD / DA_DP_OVERLAP (0 | 1) override the defaults.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class BucketedAllReducer:

    def __init__(self, flat_grad: torch.Tensor, bucket_elems: int = 64 * 1024 * 1024, group=None, align: int = 64,
                 collective: Optional[str] = None, payload: Optional[str] = None, overlap: Optional[bool] = None):
        if flat_grad.dim() != 1 or not flat_grad.is_contiguous():
            raise ValueError('flat_grad must be a contiguous 1-D tensor')
        self.flat = flat_grad
        self.bucket = int(bucket_elems)
        self.align = align
        self.group = group
        import os
        # DA_DP_FORCE=1: run the exchange even in a 1-rank group (lets a 1-GPU box execute the RCCL calls themselves)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get('DA_DP_FORCE') == '1')
        self.hi = flat_grad.numel()
        self.handles: List = []
        self.launched: List = []  # (lo, hi) ranges, for tests
        self.stream: Optional[torch.cuda.Stream] = torch.cuda.Stream() if flat_grad.is_cuda else None
        # optional per-bucket continuation, run stream-ordered behind the bucket's all-reduce (e.g. its AdamW slice)
        self.on_bucket = None
        self.collective = collective or os.environ.get('DA_DP_COLLECTIVE', 'allreduce')
        self.payload = payload or os.environ.get('DA_DP_PAYLOAD', 'fp32')
        if self.collective not in ('allreduce', 'rs_ag') or self.payload not in ('fp32', 'bf16'):
            raise ValueError(f'collective must be allreduce|rs_ag and payload fp32|bf16, got {self.collective}, {self.payload}')
        self.overlap = (os.environ.get('DA_DP_OVERLAP', '1') != '0') if overlap is None else bool(overlap)
        self._stage = {}
        # CUs left to the collective WHILE one is in flight (da_set_option("reserve_cus")): set when the first bucket of a
        # step is handed to RCCL, cleared in flush().  Kernels are enqueued in order, so everything enqueued between the two
        # - the part of backward that runs beside the exchange - sizes its one-round grids to #CUs - R.  Held for the whole
        # step it measured -7 % at one GPU (tile counts are multiples of 256, so 248 workgroups walk 33 rounds' worth in 34);
        # scoped to the overlap window it costs that on ~1/3 of the step.  The trainer sets it (DA_DP_RESERVE_CUS).
        self.reserve_cus = 0
        self._reserved = False

    def _exchange(self, view: torch.Tensor):
        """Sum ``view`` over the ranks, in place, stream-ordered on the current stream (RCCL: ``wait()`` only makes
        the current stream wait; gloo: host-blocking)."""
        world = dist.get_world_size(self.group)
        n = view.numel()
        if self.collective == 'allreduce' and self.payload == 'fp32':
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            return
        rank = dist.get_rank(self.group)
        if self.payload == 'fp32' and n % world == 0 and dist.get_backend(self.group) == 'nccl':
            # rs_ag on the gradient where it lies: RCCL's in-place forms (reduce-scatter output = this rank's slice of the
            # input, all-gather input = this rank's slice of the output) - no staging copy, no per-step allocation
            shard = view.view(world, n // world)[rank]
            dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            dist.all_gather_into_tensor(view, shard, group=self.group, async_op=True).wait()
            return
        npad = -(-n // world) * world
        dt = torch.bfloat16 if self.payload == 'bf16' else view.dtype
        key = (npad, dt)
        bufs = self._stage.get(key)
        if bufs is None:   # one staging buffer (+ shard) per distinct bucket size (the bucket sequence repeats every step)
            bufs = self._stage[key] = (torch.zeros(npad, device=view.device, dtype=dt),
                                       torch.zeros(npad // world, device=view.device, dtype=dt))
        stage, shard = bufs
        stage[:n].copy_(view)
        if self.collective == 'allreduce':
            dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
        else:
            dist.reduce_scatter_tensor(shard, stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            dist.all_gather_into_tensor(stage, shard, group=self.group, async_op=True).wait()
        view.copy_(stage[:n])

    def _set_reserve(self, r: int):
        from . import ops
        ops.set_option('reserve_cus', int(r))
        self._reserved = r > 0

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if self.enabled else 1

    def begin(self):
        self.hi = self.flat.numel()
        self.handles.clear()
        self.launched.clear()

    def _launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if not self.enabled and self.on_bucket is None:
            return
        view = self.flat[lo:hi]
        if self.enabled and self.reserve_cus and not self._reserved and self.overlap:
            self._set_reserve(self.reserve_cus)
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                if self.enabled:
                    self._exchange(view)     # ordered on the side stream; the continuation follows on it
                if self.on_bucket is not None:
                    self.on_bucket(lo, hi)
        else:
            if self.enabled:
                self._exchange(view)
            if self.on_bucket is not None:
                self.on_bucket(lo, hi)

    def ready(self, lo: int):
        """Every gradient at flat offsets >= lo is final."""
        lo = (lo // self.align) * self.align
        if not self.overlap:
            return            # everything goes out in flush(), bucket by bucket
        if self.hi - lo >= self.bucket:
            self._launch(lo, self.hi)
            self.hi = lo

    def abort(self):
        """Leave a step that raised: give the reserved CUs back and forget the unlaunched range."""
        self.on_bucket = None
        self.hi = 0
        if self._reserved:
            self._set_reserve(0)

    def flush(self):
        """Launch what is left ([0, hi)) and make the current stream wait for every bucket."""
        if not self.overlap:      # same bucket boundaries as the overlapped schedule would have produced
            while self.hi > self.bucket:
                lo = ((self.hi - self.bucket) // self.align) * self.align
                self._launch(lo, self.hi)
                self.hi = lo
        try:
            self._launch(0, self.hi)
        finally:
            self.hi = 0
            if self._reserved:
                self._set_reserve(0)
        for h in self.handles:
            h.wait()
        if self.stream is not None and (self.enabled or self.on_bucket is not None):
            torch.cuda.current_stream().wait_stream(self.stream)
        self.handles.clear()


def init_distributed_from_env(device_index: Optional[int] = None):
    """torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*) -> process group; RCCL when a GPU is present."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        # one rank per GPU; (DA_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path on a 1-GPU box)
        n = torch.cuda.device_count()
        torch.cuda.set_device((local if device_index is None else device_index) % max(n, 1))
    if (world > 1 or os.environ.get('DA_DP_FORCE') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('DA_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
