"""CPU replay of the COMPILED op list of the backbone against the CPU oracle (no GPU, no HIP kernel).

What it checks is the host logic the GPU tests can only check together with the kernels: the op list
`HighResolutionNet._compile` hands to `shapy_hrnet_run` (the ctypes `ShapyOp` array, exactly what the
executor in csrc/hrnet_ops.hip reads), the packed weight blob and the packed per-image workspace -- buffer
offsets, row strides and channel offsets, residual aliasing (in-place accumulation of the fuse layers),
upsample terms, the concat buffer of the head.  Every op is restated with torch CPU ops on
ONE flat float32 workspace laid out as the executor addresses it (buffer = ws + off * B), in plan order
(= enqueue order: one legal schedule of the event-driven plan), always from the op's direct-convolution
weights (`wgt_off`; a Winograd layer's transformed filters are the kernel tests' business).
"""
import os.path as osp
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)


def replay(eng, x):
    """eng: HighResolutionNet._compile(...) on the CPU; x [B,3,H,W] float32.  Returns features [B, C]."""
    from shapy_amd import _lib
    B = x.shape[0]
    blob = eng['weights'].numpy().view(np.float32)
    ws = torch.zeros(int(eng['ws_per_img']) * B, dtype=torch.float32)
    ws += float('nan')                                  # reading what nobody wrote poisons the result

    def rows(off, n_pix, ld):
        return ws[off * B: off * B + B * n_pix * ld].view(B * n_pix, ld)

    def wts(off, n):
        return torch.from_numpy(blob[off: off + n].copy())

    def up(t, f):                                       # [B,H,W,C] nearest upsample
        return t.repeat_interleave(f, 1).repeat_interleave(f, 2)
    feats = None
    for a in eng['ops'][:eng['n_ops']]:
        if a.type == _lib.OP_STEM:
            assert a.in_off == -2
            w = wts(a.wgt_off, 64 * 27).view(64, 3, 3, 3).permute(0, 3, 1, 2)       # [n][kh][kw][c]
            y = F.relu(F.conv2d(x, w, wts(a.bias_off, 64), stride=2, padding=1))
            rows(a.out_off, a.Ho * a.Wo, a.out_ld)[:, a.out_coff:a.out_coff + 64] = \
                y.permute(0, 2, 3, 1).reshape(-1, 64)
        elif a.type == _lib.OP_CONV:
            xin = rows(a.in_off, a.Hi * a.Wi, a.in_ld)[:, :a.Cin].reshape(B, a.Hi, a.Wi, a.Cin)
            w = wts(a.wgt_off, a.Cout * a.ksize * a.ksize * a.Cin).view(a.Cout, a.ksize, a.ksize, a.Cin)
            y = F.conv2d(xin.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), wts(a.bias_off, a.Cout),
                         stride=a.stride, padding=a.pad).permute(0, 2, 3, 1)
            assert y.shape[1:3] == (a.Ho, a.Wo)
            if a.ups > 1:
                y = up(y, a.ups)
            n_pix = a.Ho * a.ups * a.Wo * a.ups
            y = y.reshape(-1, a.Cout)
            if a.res_off >= 0:
                y = y + rows(a.res_off, n_pix, a.res_ld)[:, a.res_coff:a.res_coff + a.Cout]
            if a.relu:
                y = F.relu(y)
            rows(a.out_off, n_pix, a.out_ld)[:, a.out_coff:a.out_coff + a.Cout] = y
        elif a.type == _lib.OP_MEANPOOL:
            feats = rows(a.in_off, a.Hi * a.Wi, a.in_ld)[:, :a.Cin].reshape(B, a.Hi * a.Wi, a.Cin).mean(1)
        else:
            raise AssertionError(f'unknown op type {a.type}')
    return feats


@pytest.fixture(scope='module')
def seeded():
    from oracle import hrnet_torch
    from shapy_amd.config import default_config
    from shapy_amd.models.backbone.hrnet import HighResolutionNet
    from shapy_amd.utils import synthetic as syn
    net = HighResolutionNet(default_config().network.smplx.backbone.hrnet).eval()
    syn.fill_module_synthetic(net, 3, prefix='backbone.', only_prefixes=('backbone.',))
    x = torch.from_numpy(syn.synthetic_images(2, 64, 17))
    sd = {'backbone.' + k: v.detach() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    with torch.no_grad():
        ref = hrnet_torch.hrnet_forward(sd, x, prefix='backbone.')
    return net, x, ref


@pytest.mark.parametrize('dag', [True, False, 'grouped'])
def test_compiled_plan_replayed_on_the_cpu_equals_the_oracle(seeded, dag):
    net, x, ref = seeded
    keep = net.dag, net.multi_stream, net.group_branches, net.conv_algo
    try:
        net.dag, net.multi_stream = bool(dag), True
        net.conv_algo = 'direct'   # (the replay reads the direct weights: no need to transform 209 filters)
        if dag == 'grouped':       # persistent launch groups per depth level (needs the F(4x4) layers)
            net.group_branches, net.conv_algo = True, 'winograd4'
        eng = net._compile(64, 64, torch.device('cpu'))
    finally:
        net.dag, net.multi_stream, net.group_branches, net.conv_algo = keep
    from shapy_amd import _lib
    n_ups = sum(1 for a in eng['ops'][:eng['n_ops']] if a.type == _lib.OP_CONV and a.ups > 1)
    # upsample terms of W48: stage 2: 1, stage 3: 4 x 3, stage 4: 3 x 6
    if dag == 'grouped':
        assert sum(1 for a in eng['ops'][:eng['n_ops']] if a.group > 1) >= 8      # (64 x 64: stage 2 only)
    assert n_ups == 31 and eng['n_ops'] == 332
    with torch.no_grad():
        feats = replay(eng, x)
    assert feats.shape == ref.shape and torch.isfinite(feats).all()
    scale = ref.abs().max().item()
    err = (feats - ref).abs().max().item()
    assert scale > 0.5 and err < 2e-5 * max(1.0, scale), (scale, err)
