"""First hardware run of the `fuse_add` plans (HighResolutionNet.fuse_add, SHAPY_OP_FUSEADD).

The path was written at the end of round 4 AFTER the round's GPU budget was spent: its plan, buffer
packing and arithmetic are checked on the CPU (tests/test_plan_replay.py, tests/test_host_cpu.py), its
kernel (csrc/hrnet_ops.hip: fuse_add_kernel) has only been compiled.  It is OFF by default; these tests
are its first execution.  They are marked xfail(strict=False) for exactly that reason -- not because a
failure is expected, but so that a kernel bug of an opt-in path that could not be run yet shows up as
XFAIL (and a working path as XPASS) instead of stopping the suite of the product path.  Remove the marker
with the first green run.  (File name: collected last.)
"""
import os
import os.path as osp

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.xfail(strict=False, reason='opt-in fuse_add path: written after the GPU budget of '
                                                     'round 4 was spent, first hardware run')]

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


@pytest.fixture(scope='module')
def backbone():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from shapy_amd.config import default_config
    from shapy_amd.models.backbone.hrnet import HighResolutionNet
    from shapy_amd.utils import synthetic as syn
    net = HighResolutionNet(default_config().network.smplx.backbone.hrnet).eval()
    syn.fill_module_synthetic(net, 0, prefix='backbone.', only_prefixes=('backbone.',))
    net = net.to('cuda')
    net.wino_guard = False                       # same launches in both runs, no calibration pass
    return net


def _features(bb, x, **opts):
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
    assert torch.equal(out[0], out[1])           # deterministic from forward to forward
    return out[0]


@pytest.mark.parametrize('form,lanes', [(1, None), (2, 'dest,dest,mixed'), (2, 'source,source,source')])
@pytest.mark.parametrize('B,size,algo,multi', [(2, 64, 'direct', True), (3, 96, 'direct', False),
                                               (3, 224, 'winograd4', True), (2, 256, 'winograd4', True),
                                               (64, 224, 'winograd4', True)])
def test_fuse_add_plan_equals_the_scatter_plan(backbone, B, size, algo, multi, form, lanes):
    """Same convolutions; form 1: same order of the additions (base + up2 + up4 + up8), form 2: the
    stride-2 terms summed apart from x_i (another association of the same sum): the features of the
    fuse_add plans equal those of the upsample-scatter plan to rounding -- 1e-5 of the feature scale."""
    from shapy_amd.utils import synthetic as syn
    x = torch.from_numpy(syn.synthetic_images(B, size, 31)).cuda()
    ref = _features(backbone, x, fuse_add=0, conv_algo=algo, multi_stream=multi)
    opts = dict(fuse_chain_lanes=lanes) if lanes else {}
    got = _features(backbone, x, fuse_add=form, conv_algo=algo, multi_stream=multi, **opts)
    plan = [e for k, e in backbone._engine.items() if k[0] == size and k[14] == form]
    n_add = 26 if form == 2 else 18
    assert any(sum(1 for o in e['plan'].ops if o['type'] == 3) == n_add for e in plan)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f'fuse_add vs scatter plan: B={B} {size}x{size} {algo}: scale {scale:.3g}, max diff {err:.2e}')
    assert scale > 0.5 and err <= 1e-5 * max(1.0, scale)


def test_fuse_add_features_vs_cpu_oracle(backbone):
    from oracle import hrnet_torch
    from shapy_amd.utils import synthetic as syn
    x = torch.from_numpy(syn.synthetic_images(2, 96, 32)).cuda()
    got = _features(backbone, x, fuse_add=2, conv_algo='direct', multi_stream=True).cpu()
    sd = {'backbone.' + k: v.detach().cpu() for k, v in backbone.state_dict().items()}
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    with torch.no_grad():
        ref = hrnet_torch.hrnet_forward(sd, x.cpu(), prefix='backbone.')
    assert (got - ref).abs().max().item() < 1e-4


def test_fuse_add_bf16_storage(backbone):
    """bf16 activations: a low-resolution term is rounded to bf16 once before it is added (the scatter
    form adds the float32 accumulator) -- a bf16-sized difference, bounded by the bf16 plan's own error."""
    from shapy_amd.utils import synthetic as syn
    x = torch.from_numpy(syn.synthetic_images(4, 224, 33)).cuda()
    ref32 = _features(backbone, x, fuse_add=0, compute_dtype='f32', conv_algo='direct', multi_stream=True)
    ref = _features(backbone, x, fuse_add=0, compute_dtype='bf16', multi_stream=True)
    got = _features(backbone, x, fuse_add=2, compute_dtype='bf16', multi_stream=True)
    e_ref = (ref.float() - ref32).abs().max().item()
    e_got = (got.float() - ref32).abs().max().item()
    print(f'bf16 error vs f32: scatter plan {e_ref:.3g}, fuse_add plan {e_got:.3g}')
    assert e_got <= 1.5 * e_ref + 1e-3
