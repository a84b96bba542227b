"""Localises errors of the F(4x4) kernel (debug aid of GPU run B, round 4): per case the error against
float64 reduced over tile position (y % 4, x % 4), channel, tile row / column and image."""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from shapy_amd import _lib  # noqa: E402
import test_gpu_parity as T  # noqa: E402

CASES = [(1, 8, 8, 16, 48, False, False), (1, 8, 8, 16, 48, False, True), (1, 8, 8, 16, 48, True, False),
         (1, 8, 8, 48, 48, False, False), (2, 12, 20, 16, 48, False, False),
         (2, 12, 20, 48, 48, True, True), (1, 14, 14, 192, 192, True, True)]
lib = _lib.load()
print('lib', _lib.LIB_PATH)
for case in CASES:
    B, H, W, Cin, Cout, use_res, relu = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / np.sqrt(9 * Cin)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, H, W, Cout, generator=g).cuda() if use_res else None
    out = T._conv_call(lib, x, w, b, res, relu, 1, 1, wino=4)
    ref = T._conv_ref(x, w, b, res, relu, 1, 1)
    e = (out.cpu().double() - ref).abs()
    tol = 2e-6 * np.sqrt(9 * Cin)
    bad = e > tol
    print(case, 'max err %.3g' % e.max().item(), 'bad fraction %.3f' % bad.float().mean().item())
    if bad.any():
        yy = torch.arange(H) % 4
        xx = torch.arange(W) % 4
        pos = torch.zeros(4, 4)
        for a in range(4):
            for c in range(4):
                pos[a, c] = bad[:, yy == a][:, :, xx == c].float().mean()
        print('  bad by (y%4, x%4):', np.round(pos.numpy(), 2).tolist())
        print('  bad by channel:', np.round(bad.float().mean(dim=(0, 1, 2)).numpy(), 2).tolist())
        print('  bad by row y:', np.round(bad.float().mean(dim=(0, 2, 3)).numpy(), 2).tolist())
        print('  bad by col x:', np.round(bad.float().mean(dim=(0, 1, 3)).numpy(), 2).tolist())
        print('  bad by image:', np.round(bad.float().mean(dim=(1, 2, 3)).numpy(), 2).tolist())
        o = out.cpu().double()
        # is the output a permutation of the reference? (channel c of the output vs channel c' of ref)
        if Cout <= 48:
            m = np.zeros((Cout, Cout))
            for c in range(Cout):
                m[c] = (o[..., c:c + 1] - ref).abs().mean(dim=(0, 1, 2)).numpy()
            print('  best matching ref channel per out channel:', m.argmin(axis=1).tolist())

# ---- where do the wrong values come from? (second part of the debug run) ----
print('=== value forensics, case (2, 12, 20, 16, 48, False, False)')
B, H, W, Cin, Cout = 2, 12, 20, 16, 48
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / np.sqrt(9 * Cin)).cuda()
b = torch.randn(Cout, generator=g).cuda()
out = T._conv_call(lib, x, w, b, None, False, 1, 1, wino=4).cpu().double()
ref = T._conv_ref(x, w, b, None, False, 1, 1)
nob = T._conv_ref(x, w, None, None, False, 1, 1)          # without bias
bad = ((out - ref).abs() > 1e-3).nonzero()
print('bad elements', len(bad))
flat = ref.reshape(-1)
for (bi, y, xx, c) in bad[:24].tolist():
    v = out[bi, y, xx, c].item()
    k = (flat - v).abs().argmin().item()
    idx = np.unravel_index(k, tuple(ref.shape))
    print(f'  out[{bi},{y},{xx},{c}] = {v:+.5f}  ref {ref[bi, y, xx, c].item():+.5f}  bias {b[c].item():+.5f} '
          f'conv-only {nob[bi, y, xx, c].item():+.5f} | nearest ref value at {tuple(int(i) for i in idx)} '
          f'({flat[k].item():+.5f})')
