"""Exact 3-way bfloat16 split of float32 weights for the bf16x6 convolution (csrc/conv_x6.hip).

Every finite float32 ``a`` is exactly ``h + m + l`` with three bfloat16 numbers (8 + 8 + 8
significand bits): ``h`` = ``a`` truncated to bfloat16, ``m`` = ``a - h`` truncated, ``l`` =
``a - h - m`` (exact).  The kernel wants the weights of one convolution as three planes per
output channel, ``[Cout, 3, Kp]`` bfloat16 with ``K = ks*ks*Cin`` zero-padded to a multiple of
32 -- 1.5x the float32 bytes, split once when the weights are packed.
"""
import numpy as np

_HI = np.uint32(0xffff0000)


def split_bf16x3(w):
    """float32 array [..., K] -> uint16 array [..., 3, Kp] (bfloat16 bit patterns)."""
    a = np.ascontiguousarray(w, dtype=np.float32)
    K = a.shape[-1]
    Kp = (K + 31) // 32 * 32
    h = (a.view(np.uint32) & _HI).view(np.float32)
    r1 = a - h                                           # exact
    m = (r1.view(np.uint32) & _HI).view(np.float32)
    r2 = r1 - m                                          # exact, <= 8 significant bits
    out = np.zeros(a.shape[:-1] + (3, Kp), np.uint16)
    for i, part in enumerate((h, m, r2)):
        out[..., i, :K] = (np.ascontiguousarray(part).view(np.uint32) >> np.uint32(16)).astype(np.uint16)
    return out


def join_bf16x3(planes, K=None):
    """Inverse of ``split_bf16x3`` (float32); used by the tests."""
    p = np.asarray(planes, np.uint16).astype(np.uint32) << np.uint32(16)
    f = p.view(np.float32)
    s = (f[..., 2, :] + f[..., 1, :]) + f[..., 0, :]
    return s[..., :K] if K is not None else s
