"""Train model - mirrors /root/reference diffusion/train.py:21-138 on the in-tree trainer.

Same config contract (yamls/hydra-yamls/SD-2-base-256.yaml): ``seed``, ``model``, ``optimizer``,
``dataset.train_dataset`` (batch split by world size, train.py:40), ``algorithms`` (the low-precision
GroupNorm/LayerNorm surgery of train.py:91-108 is a no-op here: the HIP norms already read/write bf16 with fp32
statistics), ``callbacks``, ``scheduler``, ``trainer``.  Evaluators / W&B loggers are out of the hot-path scope."""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.distributed as dist

from . import hydra_lite as hydra
from .parallel import init_distributed_from_env


def seed_all(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def train(config) -> None:
    rank, local, world = init_distributed_from_env()
    seed_all(int(config['seed']) + rank)

    model = hydra.instantiate(config.model)
    optimizer = hydra.instantiate(config.optimizer, params=model.parameters(), unet=model.unet)

    train_dataloader = hydra.instantiate(
        config.dataset.train_dataset,
        batch_size=config.dataset.train_batch_size // world,
        _recursive_=False,
    )

    # evaluation data (reference train.py:44-63): a plain `dataset.eval_dataset` is built like the train loader with the
    # per-rank share of `eval_batch_size`; `dataset.evaluators` (FID / CLIP-score Evaluator objects) are out of scope
    eval_dataloader = None
    ds = config.dataset
    if 'eval_dataset' in ds and ds.eval_dataset and '_target_' in ds.eval_dataset:
        eval_dataloader = hydra.instantiate(ds.eval_dataset, batch_size=int(ds.get('eval_batch_size', ds.train_batch_size)) // world,
                                            _recursive_=False)

    callbacks = []
    if 'callbacks' in config and config.callbacks:
        for _, call_conf in config.callbacks.items():
            if call_conf and '_target_' in call_conf:
                callbacks.append(hydra.instantiate(call_conf))
    # algorithms: low_precision_groupnorm / low_precision_layernorm are inherent to the kernels (train.py:91-108);
    # objects with a _target_ (EMA, SD-2-base-512.yaml:8-13) are instantiated like the reference does (train.py:86-90)
    algorithms = []
    if 'algorithms' in config and config.algorithms:
        for _, ag_conf in config.algorithms.items():
            if ag_conf and '_target_' in ag_conf:
                algorithms.append(hydra.instantiate(ag_conf))
    scheduler = hydra.instantiate(config.scheduler) if config.get('scheduler') else None

    trainer_conf = dict(config.trainer)
    trainer = hydra.instantiate(
        trainer_conf,
        train_dataloader=train_dataloader,
        eval_dataloader=eval_dataloader,
        optimizers=optimizer,
        model=model,
        loggers=[],
        algorithms=algorithms,
        schedulers=scheduler,
        callbacks=callbacks,
    )
    if config.get('eval_first', True):
        trainer.eval(subset_num_batches=trainer_conf.get('eval_subset_num_batches'))
    trainer.fit()
    if dist.is_initialized():
        dist.barrier()
    return trainer
