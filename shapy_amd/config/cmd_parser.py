"""``--exp-cfg <yaml...> --exp-opts k=v ...`` (reference ``config/cmd_parser.py:12-49``)."""
import argparse

from .defaults import default_config
from .node import ConfigNode


def build_parser(description='Human body regressor'):
    parser = argparse.ArgumentParser(
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, description=description)
    parser.add_argument('--exp-cfg', type=str, dest='exp_cfgs', required=True, nargs='+',
                        help='The configuration of the experiment')
    parser.add_argument('--exp-opts', default=[], dest='exp_opts', nargs='*',
                        help='Dot-list overrides, e.g. network.smplx.num_stages=3')
    parser.add_argument('--local_rank', default=0, type=int, help='ranking within the nodes')
    parser.add_argument('--num-gpus', dest='num_gpus', default=1, type=int,
                        help='Number of gpus')
    parser.add_argument('--backend', dest='backend', default='nccl', type=str,
                        choices=['nccl', 'gloo'],
                        help='torch.distributed backend ("nccl" is RCCL on ROCm)')
    return parser


def merge_config(exp_cfgs=(), exp_opts=()):
    """defaults (+) YAML files in order (+) dot-list, later wins."""
    cfg = default_config()
    for path in exp_cfgs:
        if path:
            cfg.merge_with(ConfigNode.load(path))
    if exp_opts:
        cfg.merge_with(ConfigNode.from_dotlist(list(exp_opts)))
    return cfg


def parse_args(argv=None):
    cmd_args = build_parser().parse_args(argv)
    cfg = merge_config(cmd_args.exp_cfgs, cmd_args.exp_opts)
    cfg.network.use_sync_bn = bool(cfg.network.use_sync_bn and cmd_args.num_gpus > 1)
    cfg.local_rank = cmd_args.local_rank
    cfg.num_gpus = cmd_args.num_gpus
    cfg.backend = cmd_args.backend
    return cfg
