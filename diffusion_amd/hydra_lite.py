"""The slice of Hydra/OmegaConf the reference's configs use (run.py:14-22, diffusion/train.py:31-128), on PyYAML:
``_target_`` dotted-path instantiation (recursive, kwargs overridable), ``${a.b}`` interpolation, ``key=value``
command-line overrides, ``_partial_`` / ``_recursive_``.  hydra-core and omegaconf are not installed here.

Targets are resolved through ``TARGET_ALIASES`` first so the reference's YAMLs run unmodified:
``diffusion.*`` -> ``diffusion_amd.*``, ``composer.Trainer`` -> the in-tree trainer, ``torch.optim.AdamW`` -> the
fused HIP AdamW (when given U-Net parameters), observability callbacks/loggers -> no-ops."""
from __future__ import annotations

import functools
import importlib
import re
from typing import Any, Dict, List

import yaml

TARGET_ALIASES = {
    'composer.Trainer': 'diffusion_amd.trainer.Trainer',
    'composer.optim.MultiStepWithWarmupScheduler': 'diffusion_amd.trainer.MultiStepWithWarmupScheduler',
    'composer.callbacks.speed_monitor.SpeedMonitor': 'diffusion_amd.trainer.SpeedMonitor',
    'composer.callbacks.SpeedMonitor': 'diffusion_amd.trainer.SpeedMonitor',
    'torchmetrics.MeanSquaredError': 'diffusion_amd.models.composer_shim.MeanSquaredError',
    'torch.optim.AdamW': 'diffusion_amd.optim.FusedAdamW',
}
TARGET_PREFIX_ALIASES = [
    ('diffusion.', 'diffusion_amd.'),
    ('composer.callbacks.', 'diffusion_amd.trainer.NoOpCallback'),
    ('composer.loggers.', 'diffusion_amd.trainer.NoOpCallback'),
    ('composer.algorithms.', 'diffusion_amd.trainer.NoOpCallback'),
]

_INTERP = re.compile(r'\$\{([^}]+)\}')


class Config(dict):
    """dict with attribute access (DictConfig-like)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lookup(root, path):
    cur = root
    for part in path.split('.'):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


def _resolve(node, root):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve(node[k], root)
        return node
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:
            return _resolve(_lookup(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root)), node)
    return node


def _parse_scalar(s: str):
    return yaml.safe_load(s)


def load_config(path: str, overrides: List[str] = ()) -> Config:
    with open(path) as f:
        cfg = _wrap(yaml.safe_load(f) or {})
    for ov in overrides:
        if '=' not in ov:
            raise ValueError(f'override {ov!r} is not key=value')
        key, val = ov.split('=', 1)
        key = key.lstrip('+')
        cur = cfg
        parts = key.split('.')
        for part in parts[:-1]:
            if part not in cur or cur[part] is None:
                cur[part] = Config()
            cur = cur[part]
        cur[parts[-1]] = _wrap(_parse_scalar(val))
    return _resolve(cfg, cfg)


def resolve_target(path: str):
    path = TARGET_ALIASES.get(path, path)
    for pre, rep in TARGET_PREFIX_ALIASES:
        if path.startswith(pre):
            path = rep if not rep.endswith('.') else rep + path[len(pre):]
            break
    path = TARGET_ALIASES.get(path, path)
    mod, _, name = path.rpartition('.')
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, _recursive_: bool = True, _partial_: bool = False, **kwargs):
    """hydra.utils.instantiate for dict nodes carrying ``_target_``."""
    if cfg is None:
        return None
    if isinstance(cfg, list):
        return [instantiate(c) if isinstance(c, dict) and '_target_' in c else c for c in cfg]
    if not isinstance(cfg, dict) or '_target_' not in cfg:
        raise ValueError('instantiate needs a mapping with _target_')
    target = resolve_target(cfg['_target_'])
    recursive = cfg.get('_recursive_', _recursive_)
    partial = cfg.get('_partial_', _partial_)
    kw: Dict[str, Any] = {}
    for k, v in cfg.items():
        if k in ('_target_', '_recursive_', '_partial_'):
            continue
        if recursive and isinstance(v, dict) and '_target_' in v:
            v = instantiate(v)
        elif recursive and isinstance(v, list):
            v = [instantiate(e) if isinstance(e, dict) and '_target_' in e else e for e in v]
        kw[k] = v
    kw.update(kwargs)
    if partial:
        return functools.partial(target, *args, **kw)
    return target(*args, **kw)
