"""Run training - same entry as /root/reference run.py:14-22 (``@hydra.main``), on hydra_lite:

    python run.py --config-path yamls/hydra-yamls --config-name SD-2-base-256.yaml [key=value ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run.py --config-path ... (one rank/GPU)
"""
import argparse
import os
import textwrap

from diffusion_amd import hydra_lite
from diffusion_amd.train import train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config-path', default=None)
    ap.add_argument('--config-name', default=None)
    ap.add_argument('overrides', nargs='*')
    a = ap.parse_args()
    if not a.config_path or not a.config_name:
        raise ValueError(
            textwrap.dedent("""\
                            Config path and name not specified!
                            Please specify these by using --config-path and --config-name, respectively."""))
    name = a.config_name if a.config_name.endswith(('.yaml', '.yml')) else a.config_name + '.yaml'
    config = hydra_lite.load_config(os.path.join(a.config_path, name), a.overrides)
    return train(config)


if __name__ == '__main__':
    main()
