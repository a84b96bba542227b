"""Runs the GPU cases of tests/test_zz_fuse_add_gpu.py in a process of its own and prints one JSON line per
case -- a fault in the not-yet-run fuse_add path must not take the pytest session of the product path with it.

    python tests/fuse_add_gpu_cases.py            # needs an MI355X
"""
import json
import os.path as osp
import sys

import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(B, size, algo, multi, form, lanes)
         for (B, size, algo, multi) in [(2, 64, 'direct', True), (3, 96, 'direct', False),
                                        (3, 224, 'winograd4', True), (2, 256, 'winograd4', True),
                                        (64, 224, 'winograd4', True)]
         for (form, lanes) in [(1, None), (2, 'dest,dest,mixed'), (2, 'source,source,source')]]


def features(bb, x, **opts):
    keep = {k: getattr(bb, k) for k in opts}
    try:
        for k, v in opts.items():
            setattr(bb, k, v)
        with torch.no_grad():
            out = [bb(x)['concat'].clone() for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        for k, v in keep.items():
            setattr(bb, k, v)
    return out[0], bool(torch.equal(out[0], out[1]))


def main():
    from oracle import hrnet_torch
    from shapy_amd.config import default_config
    from shapy_amd.models.backbone.hrnet import HighResolutionNet
    from shapy_amd.utils import synthetic as syn
    bb = HighResolutionNet(default_config().network.smplx.backbone.hrnet).eval()
    syn.fill_module_synthetic(bb, 0, prefix='backbone.', only_prefixes=('backbone.',))
    bb = bb.to('cuda')
    bb.wino_guard = False                        # same launches in both runs, no calibration pass

    def emit(**kw):
        print(json.dumps(kw), flush=True)
    refs = {}
    for (B, size, algo, multi, form, lanes) in CASES:
        x = torch.from_numpy(syn.synthetic_images(B, size, 31)).cuda()
        key = (B, size, algo, multi)
        if key not in refs:
            refs[key] = features(bb, x, fuse_add=0, conv_algo=algo, multi_stream=multi)[0]
        ref = refs[key]
        opts = dict(fuse_chain_lanes=lanes) if lanes else {}
        got, same = features(bb, x, fuse_add=form, conv_algo=algo, multi_stream=multi, **opts)
        n_add = max([sum(1 for o in e['plan'].ops if o['type'] == 3) for k, e in bb._engine.items()
                     if k[0] == size and k[14] == form] or [0])
        emit(case='plans', B=B, size=size, algo=algo, multi=multi, form=form, lanes=lanes, n_add=n_add,
             deterministic=same, scale=ref.abs().max().item(), err=(got - ref).abs().max().item())
    # against the CPU oracle
    x = torch.from_numpy(syn.synthetic_images(2, 96, 32)).cuda()
    got = features(bb, x, fuse_add=2, conv_algo='direct', multi_stream=True)[0].cpu()
    sd = {'backbone.' + k: v.detach().cpu() for k, v in bb.state_dict().items()}
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    with torch.no_grad():
        ref = hrnet_torch.hrnet_forward(sd, x.cpu(), prefix='backbone.')
    emit(case='oracle', err=(got - ref).abs().max().item())
    # bf16 storage
    x = torch.from_numpy(syn.synthetic_images(4, 224, 33)).cuda()
    ref32 = features(bb, x, fuse_add=0, compute_dtype='f32', conv_algo='direct', multi_stream=True)[0]
    ref = features(bb, x, fuse_add=0, compute_dtype='bf16', multi_stream=True)[0]
    for form in (1, 2):
        got = features(bb, x, fuse_add=form, compute_dtype='bf16', multi_stream=True)[0]
        emit(case='bf16', form=form, e_ref=(ref.float() - ref32).abs().max().item(),
             e_got=(got.float() - ref32).abs().max().item())


if __name__ == '__main__':
    main()
