"""Child process of tests/test_dp_gpu.py (not collected by pytest): one optimizer step of the tiny-width model on this
rank's slice of a fixed global batch, through Trainer.train_batch (bucketed gradient all-reduce + fused AdamW).

  single process :  python tests/dp_worker.py OUT.pt
  N ranks        :  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tests/dp_worker.py OUT.pt
                    (DA_DIST_BACKEND=gloo lets the ranks share one GPU; DA_DP_COLLECTIVE / DA_DP_PAYLOAD select the exchange)
Rank 0 writes {'grad': reduced flat gradient, 'before': master weights before the step, 'after': ... after}."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_amd  # noqa: E402,F401  (sets HSA_ENABLE_IPC_MODE_LEGACY=0 before the HIP runtime starts)
import torch  # noqa: E402
import torch.distributed  # noqa: E402

GLOBAL_BATCH, S, CTX = 8, 16, 128


def main():
    out = sys.argv[1]
    from diffusion_amd.parallel import init_distributed_from_env
    rank, local, world = init_distributed_from_env(device_index=0)
    dev = torch.device('cuda', torch.cuda.current_device())
    from diffusion_amd.models.models import stable_diffusion_2
    from diffusion_amd.optim import FusedAdamW
    from diffusion_amd.trainer import Trainer
    model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=True, fsdp=False, seed=3)
    opt = FusedAdamW(lr=1e-3, weight_decay=0.01, unet=model.unet)
    tr = Trainer(model, train_dataloader=None, optimizers=opt, max_duration='1ba')
    tr.reducer.bucket = 1_000_000          # several buckets at tiny width
    g = torch.Generator().manual_seed(11)
    full = {'image_latents': torch.randn(GLOBAL_BATCH, 4, S, S, generator=g).half(),
            'caption_latents': torch.randn(GLOBAL_BATCH, 77, CTX, generator=g).half(),
            '_noise': torch.randn(GLOBAL_BATCH, 4, S, S, generator=g),
            '_timesteps': torch.randint(0, 1000, (GLOBAL_BATCH,), generator=g)}
    per = GLOBAL_BATCH // world
    mine = {k: v[rank * per:(rank + 1) * per].to(dev) for k, v in full.items()}
    before = model.unet.master.detach().cpu().clone()
    loss = tr.train_batch(mine)
    torch.cuda.synchronize()
    if rank == 0:
        scale = 1.0 / world   # the exchanged gradient is the SUM over ranks; 1/world is folded into AdamW
        torch.save({'grad': (model.unet.grad.detach().cpu() * scale), 'before': before,
                    'after': model.unet.master.detach().cpu().clone(), 'loss': float(loss.item()),
                    'buckets': len(tr.reducer.launched), 'world': world, 'reducer_enabled': tr.reducer.enabled,
                    'reserve_cus': tr.reserve_cus, 'sliced': tr.sliced_optimizer,
                    'backend': (torch.distributed.get_backend() if torch.distributed.is_initialized() else None)}, out)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
