"""Host side of the Winograd F(2x2, 3x3) convolution path (csrc/conv_wino.hip).

    Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          (Lavin & Gray, minimal filtering)

``transform_filters`` turns folded OHWI conv weights into the layout the kernel streams as its
MFMA B operand: ``U[p = 4 i + j][Cin / 16][Cout][16]`` float32 with ``U[i][j] = (G g G^T)[i][j]``,
computed in float64 (the halves of G are exact there) and rounded once.
"""
import numpy as np

G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
BT = np.array([[1.0, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
AT = np.array([[1.0, 1, 1, 0], [0, 1, -1, -1]])


def eligible(ksize, stride, pad, cin, cout, ups=1):
    """Layers conv_wino.hip takes (ShapyConv.wgt_wino): 3x3 / stride 1 / pad 1, 16-channel
    K chunks, 48- or 64-channel N tiles (F(2x2): conv_wino.hip)."""
    return (ksize == 3 and stride == 1 and pad == 1 and ups == 1 and cin % 16 == 0
            and (cout % 48 == 0 or cout % 64 == 0))


def transform_filters(w_ohwi):
    """[Cout, 3, 3, Cin] -> float32 [16, Cin // 16, Cout, 16]."""
    w = np.asarray(w_ohwi, np.float64)
    cout, kh, kw, cin = w.shape
    assert (kh, kw) == (3, 3) and cin % 16 == 0, w.shape
    u = np.einsum('ia,oabc,jb->ijoc', G, w, G)                    # [4,4,Cout,Cin]
    u = u.reshape(16, cout, cin // 16, 16).transpose(0, 2, 1, 3)   # [16, Cin/16, Cout, 16]
    return np.ascontiguousarray(u, dtype=np.float32)


def conv_reference(x_nhwc, u, bias=None):
    """NumPy float32 restatement of the kernel's arithmetic (transform order included) for
    tests: x [B,H,W,Cin] -> [B,H,W,Cout]; u from ``transform_filters``."""
    f32 = np.float32
    x = np.asarray(x_nhwc, f32)
    B, H, W, C = x.shape
    cout = u.shape[2]
    TH, TW = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((B, 2 * TH + 2, 2 * TW + 2, C), f32)
    xp[:, 1:H + 1, 1:W + 1] = x
    # patches d[b,ty,tx,i,k,c]
    d = np.stack([np.stack([xp[:, i:i + 2 * TH:2, k:k + 2 * TW:2] for k in range(4)], axis=3)
                  for i in range(4)], axis=3)
    t = np.stack([d[..., 0, :] - d[..., 2, :], d[..., 1, :] + d[..., 2, :],
                  d[..., 2, :] - d[..., 1, :], d[..., 1, :] - d[..., 3, :]], axis=4)   # d B : [..,i,j,c]
    v = np.stack([t[:, :, :, 0] - t[:, :, :, 2], t[:, :, :, 1] + t[:, :, :, 2],
                  t[:, :, :, 2] - t[:, :, :, 1], t[:, :, :, 1] - t[:, :, :, 3]], axis=3)   # B^T (d B)
    uu = u.transpose(0, 2, 1, 3).reshape(4, 4, cout, C)            # [i,j,Cout,Cin]
    m = np.einsum('byxijc,ijoc->byxijo', v.astype(np.float64), uu.astype(np.float64)).astype(f32)
    tt0 = (m[..., 0, :] + m[..., 1, :]) + m[..., 2, :]             # M A : [..,i,o]
    tt1 = (m[..., 1, :] - m[..., 2, :]) - m[..., 3, :]
    y = np.zeros((B, 2 * TH, 2 * TW, cout), f32)
    for bb, tt in enumerate((tt0, tt1)):
        y[:, 0::2, bb::2] = (tt[..., 0, :] + tt[..., 1, :]) + tt[..., 2, :]
        y[:, 1::2, bb::2] = (tt[..., 1, :] - tt[..., 2, :]) - tt[..., 3, :]
    y = y[:, :H, :W]
    if bias is not None:
        y = y + np.asarray(bias, f32)
    return y.astype(f32)


# ---- F(4x4, 3x3): csrc/conv_wino4.hip --------------------------------------------------------
# Interpolation points {0, 1, -1, 2, -2, inf} (Lavin & Gray): 36 multiplies per 4x4 output tile.
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
               [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.0]])
BT4 = np.array([[4.0, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]])
AT4 = np.array([[1.0, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0],
                [0, 1, -1, 8, -8, 1]])


def eligible4(ksize, stride, pad, cin, cout, ups=1):
    """Layers conv_wino4.hip takes: 3x3 / stride 1 / pad 1, 16-channel K chunks, 48-channel N
    tiles (four waves, each 27 of the 108 position x channel-group items of a chunk)."""
    return (ksize == 3 and stride == 1 and pad == 1 and ups == 1 and cin % 16 == 0
            and cout % 48 == 0)


def transform_filters4(w_ohwi):
    """[Cout, 3, 3, Cin] -> float32 [36, Cin // 16, Cout, 16], U[6 i + j] = (G g G^T)[i][j],
    computed in float64 and rounded once."""
    w = np.asarray(w_ohwi, np.float64)
    cout, kh, kw, cin = w.shape
    assert (kh, kw) == (3, 3) and cin % 16 == 0, w.shape
    u = np.einsum('ia,oabc,jb->ijoc', G4, w, G4)                   # [6,6,Cout,Cin]
    u = u.reshape(36, cout, cin // 16, 16).transpose(0, 2, 1, 3)   # [36, Cin/16, Cout, 16]
    return np.ascontiguousarray(u, dtype=np.float32)


def conv_reference4(x_nhwc, u, bias=None):
    """NumPy restatement of the F(4x4,3x3) path for tests: x [B,H,W,Cin] -> [B,H,W,Cout]; u from
    ``transform_filters4``.  float32 transforms, float64 accumulation of the products."""
    f32 = np.float32
    x = np.asarray(x_nhwc, f32)
    B, H, W, C = x.shape
    cout = u.shape[2]
    TH, TW = (H + 3) // 4, (W + 3) // 4
    xp = np.zeros((B, 4 * TH + 2, 4 * TW + 2, C), f32)
    xp[:, 1:H + 1, 1:W + 1] = x
    d = np.stack([np.stack([xp[:, i:i + 4 * TH:4, k:k + 4 * TW:4] for k in range(6)], axis=3)
                  for i in range(6)], axis=3)                       # [b,ty,tx,i,k,c]
    bt = BT4.astype(f32)
    t = np.einsum('jk,byxikc->byxijc', bt, d).astype(f32)          # d B   (along x)
    v = np.einsum('li,byxijc->byxljc', bt, t).astype(f32)          # B^T (d B)
    uu = u.transpose(0, 2, 1, 3).reshape(6, 6, cout, C)
    m = np.einsum('byxijc,ijoc->byxijo', v.astype(np.float64), uu.astype(np.float64)).astype(f32)
    at = AT4.astype(f32)
    s = np.einsum('qj,byxijo->byxiqo', at, m).astype(f32)          # M A   (along x)
    yt = np.einsum('pi,byxiqo->byxpqo', at, s).astype(f32)         # A^T (M A)
    y = yt.transpose(0, 1, 3, 2, 4, 5).reshape(B, 4 * TH, 4 * TW, cout)[:, :H, :W]
    if bias is not None:
        y = y + np.asarray(bias, f32)
    return y.astype(f32)
