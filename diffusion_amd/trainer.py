"""Minimal trainer with the slice of ``composer.Trainer`` the reference configures
(yamls/hydra-yamls/SD-2-base-256.yaml:82-96, built at diffusion/train.py:118-128, run at :130-138):
``max_duration`` in batches, ``device_train_microbatch_size`` gradient accumulation, LR schedule, callbacks
(SpeedMonitor = the source of the README's images/sec), checkpoint save/load.  Precision is bf16 inside the
kernels (the reference's ``amp_fp16`` + GradScaler has no counterpart: bf16 needs no loss scaling)."""
from __future__ import annotations

import os
import time
from typing import Any, Iterable, List, Optional

import torch
import torch.distributed as dist

from .optim import FusedAdamW
from .parallel import BucketedAllReducer


def _parse_time(s, default_unit='ba'):
    if isinstance(s, int):
        return s, default_unit
    s = str(s)
    for unit in ('ba', 'ep', 'sp', 'dur'):
        if s.endswith(unit):
            return float(s[:-len(unit)]) if unit == 'dur' else int(s[:-len(unit)]), unit
    return int(s), default_unit


class MultiStepWithWarmupScheduler:
    """composer.optim.MultiStepWithWarmupScheduler: linear warm-up over t_warmup then x gamma at each milestone."""

    def __init__(self, t_warmup='0ba', milestones=(), gamma: float = 0.1, scale_warmup: bool = False):
        self.t_warmup = _parse_time(t_warmup)
        self.milestones = [_parse_time(m) for m in milestones]
        self.gamma = gamma

    def __call__(self, batch_idx: int, batches_per_epoch: Optional[int] = None) -> float:
        w, unit = self.t_warmup
        if unit == 'ep':
            w = w * (batches_per_epoch or 0)
        if w and batch_idx < w:
            return batch_idx / w
        f = 1.0
        for m, unit in self.milestones:
            mb = m * batches_per_epoch if (unit == 'ep' and batches_per_epoch) else (m if unit == 'ba' else None)
            if mb is not None and batch_idx >= mb:
                f *= self.gamma
        return f


class Callback:

    def batch_end(self, trainer):
        pass

    def fit_end(self, trainer):
        pass


class SpeedMonitor(Callback):
    """samples/sec over a sliding window of batches (composer.callbacks.SpeedMonitor, window_size=10 in the YAML)."""

    def __init__(self, window_size: int = 10, **kw):
        self.window = window_size
        self.times: List[float] = []
        self.throughput = None

    def batch_end(self, trainer):
        torch.cuda.synchronize()
        self.times.append(time.perf_counter())
        if len(self.times) > self.window + 1:
            self.times.pop(0)
        if len(self.times) > 1:
            dt = self.times[-1] - self.times[0]
            self.throughput = (len(self.times) - 1) * trainer.global_batch_size / dt
            trainer.log({'throughput/samples_per_sec': self.throughput})


class NoOpCallback(Callback):
    """LRMonitor / MemoryMonitor / RuntimeEstimator / OptimizerMonitor / loggers: observability, out of scope."""

    def __init__(self, *a, **kw):
        pass


class Trainer:

    def __init__(self, model, train_dataloader: Iterable, optimizers=None, max_duration='1ba',
                 device_train_microbatch_size: Optional[int] = None, schedulers=None, callbacks=None, loggers=None,
                 algorithms=None, eval_dataloader=None, eval_interval=None, device='gpu', run_name=None, seed=None,
                 scale_schedule_ratio: float = 1.0, save_folder=None, save_interval=None, save_overwrite=True,
                 autoresume=False, load_path=None, fsdp_config=None, precision=None, log_every: int = 10,
                 use_graphs=False, eval_subset_num_batches=None, **unused):
        self.model = model
        self.dataloader = train_dataloader
        self.optimizer: FusedAdamW = optimizers
        self.max_batches, unit = _parse_time(max_duration)
        if unit != 'ba':
            raise ValueError('max_duration must be given in batches (e.g. 550000ba)')
        # 'auto' (Composer's spelling) or None: as many images per pass as the card's memory holds - large M fills the
        # 256-CU tile grids (1390 vs 368 images/s at microbatch 256 vs 16) - resolved per batch by auto_microbatch()
        self.microbatch = None if device_train_microbatch_size in (None, 'auto') else int(device_train_microbatch_size)
        self.scheduler = schedulers
        self.callbacks: List[Callback] = [c for c in (callbacks or []) if isinstance(c, Callback)]
        self.algorithms = [a for a in (algorithms or []) if hasattr(a, 'before_optimizer_step')]
        self.eval_dataloader = eval_dataloader
        self.eval_interval = _parse_time(eval_interval)[0] if eval_interval else None
        self.eval_subset_num_batches = eval_subset_num_batches
        self.save_folder = save_folder
        self.save_interval = _parse_time(save_interval)[0] if save_interval else None
        self.batch_idx = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.reducer = BucketedAllReducer(model.unet.grad)
        # AdamW slices issued behind each gradient bucket on the reducer's side stream (north_star: "all-reduce ...
        # overlapped with the optimizer step"; the reference gets the overlap from FSDP's sharded optimizer step,
        # SD-2-base-256.yaml:95-96).  On by default when there IS an exchange to hide (world > 1); at world 1 it measured
        # neutral (196.4 vs 196.6 ms/step: the slices only take CUs from the backward GEMMs), so one launch.  DA_SLICED_ADAMW=0/1.
        env = os.environ.get('DA_SLICED_ADAMW')
        self.sliced_optimizer = self.reducer.enabled if env is None else (env == '1')
        # Multi-GPU: leave R CUs to the RCCL channels of the overlapping all-reduce, so grids sized to one round of the
        # chip (persistent GEMM tile walks, weight-gradient pixel splits) do not spill into a second round when a
        # collective holds a CU.  Applied by the reducer only while buckets are in flight (parallel.py); priced in
        # DESIGN.md section 6.  DA_DP_RESERVE_CUS=R overrides (0 = whole chip); DA_DP_RESERVE_ALWAYS=1 holds it all step.
        self.reserve_cus = int(os.environ.get('DA_DP_RESERVE_CUS', '8' if self.reducer.enabled else '0'))
        self.reducer.reserve_cus = self.reserve_cus
        if os.environ.get('DA_DP_RESERVE_ALWAYS') == '1':
            from . import ops
            ops.set_option('reserve_cus', self.reserve_cus)
            self.reducer.reserve_cus = 0
        self.base_lr = self.optimizer.param_groups[0]['lr']
        self.global_batch_size = None
        self.logs: List[dict] = []
        self.log_every = log_every
        self._auto_mb = {}
        # hipGraph replay of whole microbatches (graph_step.py): True / False, or 'auto' = when a step has several small
        # microbatches.  Off by default: measured on MI355X the step is GPU-bound even at the reference YAML's microbatch
        # of 16 (372.4 images/s eager vs 372.4 replayed: the small-M kernels, not the ~1,700 launches, set the time), so
        # replay buys nothing there and costs a private memory pool per captured signature.  DA_GRAPH=0/1 overrides.
        env = os.environ.get('DA_GRAPH')
        use_graphs = use_graphs if env is None else (env == '1')
        # normalised to True / False / 'auto' (a YAML's 0 / 1 / 'false' must not fall through to 'auto')
        if isinstance(use_graphs, str):
            use_graphs = 'auto' if use_graphs.lower() == 'auto' else use_graphs.lower() in ('1', 'true', 'yes', 'on')
        self.use_graphs = use_graphs if use_graphs == 'auto' else bool(use_graphs)
        self._graph_cache = None
        # resume: explicit load_path, or (autoresume) the newest checkpoint of this rank-0 run in save_folder
        self.all_algorithms = list(algorithms or [])
        self._skip_batches = 0
        if load_path:
            self.load_checkpoint(load_path)
        elif autoresume:
            if not save_folder:
                raise ValueError('autoresume=True needs save_folder')
            # rank 0 picks the checkpoint and tells everybody: only rank 0 writes ba*-rank0.pt, and a rank that globbed a
            # stale or un-shared listing by itself would silently start from batch 0 next to resumed peers
            latest = self.latest_checkpoint(save_folder) if self.rank == 0 else None
            if self.world > 1:
                box = [latest]
                dist.broadcast_object_list(box, src=0)
                latest = box[0]
            if latest:
                self.load_checkpoint(latest)
        if self.world > 1:   # every replica must continue from the same batch (a silent mismatch diverges, then hangs)
            lo = torch.tensor([self.batch_idx], dtype=torch.int64, device=model.unet.device_)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if int(lo.item()) != int(hi.item()):
                raise RuntimeError(f'ranks disagree on the resumed batch index ({int(lo.item())} .. {int(hi.item())}): '
                                   'the checkpoint must be readable on every rank')

    # saved-activation bytes per image of one forward at latent side S (bf16 activations kept for backward, measured:
    # ~60 GB for 256 images at 32^2 -> 0.235 GB/image; the count scales with the pixel count)
    ACT_GB_PER_IMAGE_AT_32 = 0.235

    def auto_microbatch(self, n: int, latent_side: int) -> int:
        """Composer's ``device_train_microbatch_size: auto`` shrinks the microbatch until it fits; here it is computed:
        the largest divisor-free chunk of the per-device batch whose saved activations fit the free HBM (with 25 %
        head-room for workspaces and the allocator), never more than the batch itself."""
        key = (n, latent_side)
        if key not in self._auto_mb:
            free, _ = torch.cuda.mem_get_info()
            free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()   # cached blocks are reusable
            per_image = self.ACT_GB_PER_IMAGE_AT_32 * (latent_side / 32.0)**2 * 2**30
            cap = max(1, int(0.75 * free / per_image))
            mb = n
            if cap < n:
                parts = -(-n // cap)
                mb = -(-n // parts)      # equal-sized microbatches
            self._auto_mb[key] = mb
            if self.rank == 0 and mb != n:
                print(f'device_train_microbatch_size=auto -> {mb} (batch {n}, latents {latent_side}x{latent_side}, '
                      f'{free / 2**30:.0f} GiB free)', flush=True)
        return self._auto_mb[key]

    def log(self, d):
        d = dict(d, batch=self.batch_idx)
        self.logs.append(d)
        if self.rank == 0 and self.batch_idx % self.log_every == 0:
            print(' '.join(f'{k}={v:.6g}' if isinstance(v, float) else f'{k}={v}' for k, v in d.items()), flush=True)

    def train_batch(self, batch) -> torch.Tensor:
        """One optimizer step: microbatched forward/backward, gradient all-reduce, fused AdamW."""
        model, unet = self.model, self.model.unet
        n = next(v.shape[0] for v in batch.values() if torch.is_tensor(v))
        self.global_batch_size = n * self.world
        mb = self.microbatch
        if mb is None:
            lat = batch.get(model.image_latents_key) if model.precomputed_latents else batch.get(model.image_key)
            side = 32 if lat is None else (lat.shape[-1] if model.precomputed_latents else lat.shape[-1] // 8)
            mb = self.auto_microbatch(n, side)
        starts = list(range(0, n, mb))
        # the first backward of the step WRITES the flat gradient (no zero fill, no read half of the read-add-writes); a
        # graph-replayed microbatch was captured with accumulate semantics and needs the zeros
        if self.use_graphs is False or self.model.unet.wgrad_stream is not None:
            unet.begin_gradient_accumulation()
        else:
            unet.zero_grad()
        total = torch.zeros((), device=unet.device_)
        self.reducer.begin()
        opt = self.optimizer
        # AdamW slices behind each gradient bucket on the side stream: measured 196.4 vs 196.6 ms/step at N=1 (and 202 ms
        # with per-kernel events) - the slices only take CUs from the backward GEMMs - so it is opt-in
        sliced = self.sliced_optimizer and hasattr(opt, 'step_range')
        try:
            for i, s in enumerate(starts):
                sub = {k: (v[s:s + mb] if torch.is_tensor(v) else v) for k, v in batch.items()}
                w = min(mb, n - s) / n
                last = i == len(starts) - 1
                graphed = self._graph_this(sub, len(starts), last)
                if not graphed:
                    outputs = model(sub)
                    loss = model.loss(outputs, sub, weight=w)
                if last:
                    # everything the optimizer step needs is known before the last backward: each gradient bucket is
                    # all-reduced and its AdamW slice issued on the side stream as soon as backward has finished it
                    if self.scheduler is not None:
                        bpe = len(self.dataloader) if hasattr(self.dataloader, '__len__') else None
                        opt.param_groups[0]['lr'] = self.base_lr * self.scheduler(self.batch_idx, bpe)
                    opt.grad_scale = 1.0 / self.world
                    for a in self.algorithms:
                        a.before_optimizer_step(self)
                    if sliced and not graphed:
                        opt.begin_step()
                        self.reducer.on_bucket = opt.step_range
                if graphed:
                    unet._grad_ready_cb = None
                    outputs, loss = self._graph_cache.step(sub, w)
                else:
                    unet._grad_ready_cb = self.reducer.ready if last else None
                    model.backward_from_loss()
                for m in model.get_metrics(is_train=True).values():
                    model.update_metric(sub, outputs, m)
                total = total + loss.detach() * w
        except BaseException:
            # an exception inside backward must not leave CUs reserved process-wide (da_set_option is global state)
            unet._grad_ready_cb = None
            self.reducer.abort()
            raise
        unet._grad_ready_cb = None
        self.reducer.flush()
        self.reducer.on_bucket = None
        opt.step()
        return total

    def _graph_this(self, sub, n_micro: int, last: bool) -> bool:
        """Replay this microbatch from a captured hipGraph?  Never the last microbatch of a multi-rank step (its backward
        overlaps the gradient exchange through host-side hooks), never with the weight-gradient side stream (its workspaces
        are re-allocated per batch shape, a captured graph would replay into freed memory)."""
        if self.use_graphs is False or (last and self.reducer.enabled) or self.model.unet.wgrad_stream is not None:
            return False
        if self._graph_cache is None:
            from .graph_step import GraphStepCache
            self._graph_cache = GraphStepCache(self.model)
        if not self._graph_cache.usable(sub):
            return False
        if self.use_graphs is True:
            return True
        lat = sub[self.model.image_latents_key]
        small = lat.shape[0] * lat.shape[-1] * lat.shape[-2] <= 64 * 1024   # <= 64 images at 32x32: launch-bound territory
        return n_micro > 1 and small

    def fit(self):
        if self._skip_batches and hasattr(self.dataloader, 'set_epoch') and len(self.dataloader):
            # resumed inside an epoch: same permutation, minus the batches already consumed (skipped at index level)
            self.dataloader.set_epoch(self.batch_idx // len(self.dataloader), skip_batches=self._skip_batches)
            self._skip_batches = 0
        it = iter(self.dataloader)
        while self.batch_idx < self.max_batches:
            try:
                batch = next(it)
            except StopIteration:
                it = iter(self.dataloader)
                batch = next(it)
            dev = self.model.unet.device_
            batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            loss = self.train_batch(batch)
            self.batch_idx += 1
            if self.batch_idx % self.log_every == 0 or self.batch_idx == self.max_batches:
                self.log({'loss/train/total': float(loss.item())})
            for c in self.callbacks:
                c.batch_end(self)
            if self.eval_interval and self.batch_idx % self.eval_interval == 0:
                self.eval()
            if self.save_folder and self.save_interval and self.batch_idx % self.save_interval == 0:
                self.save_checkpoint(os.path.join(self.save_folder, f'ba{self.batch_idx}-rank{self.rank}.pt'))
        for c in self.callbacks:
            c.fit_end(self)

    @torch.no_grad()
    def eval(self, subset_num_batches=None):
        """``composer.Trainer.eval`` for the slice the reference drives (diffusion/train.py:118-136, `eval_first`, the trainer
        block's `eval_interval` / `eval_subset_num_batches`): every batch of ``eval_dataloader`` goes through
        ``model.eval_forward`` and every validation metric through ``model.update_metric`` (stable_diffusion.py:189-257); the
        metric states are summed over the ranks (squared-error sum and count, as torchmetrics' dist_reduce_fx="sum" does) and
        logged as ``metrics/eval/<name>``.  Image-generation metrics (FID / CLIP score: val_guidance_scales) need the
        evaluation datasets and Inception / CLIP weights that are out of scope; with no eval dataloader this returns {}."""
        if self.eval_dataloader is None:
            return {}
        model = self.model
        metrics = model.get_metrics(is_train=False)
        for m in metrics.values():
            m.reset()
        n = subset_num_batches if subset_num_batches is not None else self.eval_subset_num_batches
        dev = model.unet.device_
        for i, batch in enumerate(self.eval_dataloader):
            if n is not None and n >= 0 and i >= n:
                break
            batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            outputs = model.eval_forward(batch)
            for m in metrics.values():
                model.update_metric(batch, outputs, m)
        out = {}
        for name, m in metrics.items():
            if self.world > 1 and getattr(m, 'sum_squared_error', None) is not None:
                st = torch.stack([m.sum_squared_error.double(), torch.tensor(float(m.total), device=dev, dtype=torch.float64)])
                dist.all_reduce(st)
                out[f'metrics/eval/{name}'] = float((st[0] / st[1]).item())
            else:
                out[f'metrics/eval/{name}'] = float(m.compute())
        self.log(out)
        return out

    # checkpoints keep the reference layout state['state']['model'] with diffusers key names under 'unet.' (what
    # diffusion/inference/inference_model.py:35-39 reads); optimizer moments, the EMA shadow and every algorithm's state
    # (Composer checkpoints algorithm state; reference EMA: algorithms/ema.py:280-336) ride along as CPU copies
    def save_checkpoint(self, path):
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        unet = self.model.unet
        algs = {type(a).__name__: a.state_dict() for a in self.all_algorithms if hasattr(a, 'state_dict')}
        # RNG streams (Composer checkpoints them too): the timestep / noise draws of stable_diffusion.py:177-179 come from
        # torch's global generators, so a resumed run continues the uninterrupted run's stream.  Rank 0's state is saved;
        # the other ranks re-derive theirs (load_checkpoint)
        rng = {'cpu': torch.get_rng_state(), 'cuda': torch.cuda.get_rng_state(unet.device_)}
        tmp = path + '.tmp'
        torch.save({'state': {'model': {f'unet.{k}': v.detach().cpu().contiguous() for k, v in unet.state_dict().items()},
                              'optimizers': self.optimizer.state_dict(), 'algorithms': algs, 'batch': self.batch_idx,
                              'rng': rng, 'world': self.world}}, tmp)
        os.replace(tmp, path)   # a killed run never leaves a half-written newest checkpoint for autoresume

    @staticmethod
    def latest_checkpoint(folder: str):
        import glob
        import re
        best = None
        for f in glob.glob(os.path.join(folder, 'ba*-rank0.pt')):
            m = re.search(r'ba(\d+)-rank0\.pt$', f)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), f)
        return best[1] if best else None

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location='cpu')
        sd = {k[len('unet.'):]: v for k, v in ck['state']['model'].items() if k.startswith('unet.')}
        self.model.unet.load_state_dict(sd)
        self.optimizer.load_state_dict(ck['state']['optimizers'])
        algs = ck['state'].get('algorithms', {})
        for a in self.all_algorithms:
            if type(a).__name__ in algs and hasattr(a, 'load_state_dict'):
                a.load_state_dict(algs[type(a).__name__])
        self.batch_idx = ck['state']['batch']
        rng = ck['state'].get('rng')
        if rng is not None:
            if self.rank == 0 and ck['state'].get('world', 1) == self.world:
                torch.set_rng_state(rng['cpu'])
                torch.cuda.set_rng_state(rng['cuda'], self.model.unet.device_)
            else:   # only rank 0's streams are in the file: a distinct, reproducible stream per (rank, batch)
                torch.manual_seed(int.from_bytes(rng['cpu'][:8].numpy().tobytes(), 'little') % (2**31) + 1000003 * self.rank
                                  + self.batch_idx)
        if hasattr(self.dataloader, 'set_epoch') and hasattr(self.dataloader, '__len__') and len(self.dataloader):
            # data position: epoch = batches // len, and the batches of that epoch already consumed are skipped by fit()
            self.dataloader.set_epoch(self.batch_idx // len(self.dataloader))
            self._skip_batches = self.batch_idx % len(self.dataloader)
