"""Thread-by-thread NumPy emulation of one F(4x4) workgroup in its one-staging-wave + three-multiplying-waves
form (rounds 2-5's conv_wino4.hip; since round 6 the form of the grouped persistent kernel csrc/conv_wino4g.hip
only -- the per-layer kernel lets all four waves multiply and stage, same LDS image, same fragment and filter
addressing, same epilogue): the staging wave's lane map and byte offsets (incl. the 0x40000000 "out of range" arithmetic of the
buffer loads), the XOR-swizzled LDS image, the multiplying waves' fragment addressing, the MFMA
16x16x4 operand / result lane layout with the "4 consecutive k per lane" trick, the filter-ring
addresses, the in-register output transform (one tile x four channels per lane) and the scalar-offset
16-byte residual / store addressing --
everything except the hardware semantics themselves -- and `emulate_workgroup4`, the same for the per-layer
kernel's four-multiplying-wave workgroup (staging lane map of all four waves, row-pair transform, 27-item split,
wave 3's exchange).  Compares every workgroup's output with a float64 direct convolution.  CPU only:

    python tools/wino4_emulate.py              # kernel indexing, several shapes
    python tools/wino4_emulate.py --network    # F(4x4,3x3) float32 numerics through all of HRNet-W48
"""
import os.path as osp
import sys

import numpy as np

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from shapy_amd.utils import winograd as wg      # noqa: E402

f32 = np.float32
BAD = 0x40000000
PSTR, LDS_V = 1024, 36 * 1024


def buf_load(mem_bytes, num_records, off, nbytes):
    """raw buffer load: zeros when the byte offset is out of range (unsigned compare)."""
    off &= 0xffffffff
    if off + nbytes > num_records:
        return np.zeros(nbytes // 4, f32)
    return mem_bytes[off // 4: off // 4 + nbytes // 4]


def bt6(d):
    a = d[4] - f32(4) * d[2]
    b = d[3] - f32(4) * d[1]
    c = d[4] - d[2]
    e = d[3] - d[1]
    return [f32(4) * d[0] - f32(5) * d[2] + d[4], a + b, a - b, c + f32(2) * e, c - f32(2) * e,
            f32(4) * d[1] - f32(5) * d[3] + d[5]]


def at6(m):
    s12, d12 = m[1] + m[2], m[1] - m[2]
    s34, d34 = m[3] + m[4], m[3] - m[4]
    return [(m[0] + s12) + s34, f32(2) * d34 + d12, f32(4) * s34 + s12, f32(8) * d34 + d12 + m[5]]


def emulate_workgroup(x, u, bias, res, relu, out, wg_m, wg_n, out_ld, out_coff, res_ld, res_coff):
    B, H, W, in_ld = x.shape
    Cin = in_ld
    Cout = u.shape[2]
    TW, TH = (W + 3) // 4, (H + 3) // 4
    T = B * TH * TW
    CC = Cin // 16
    NW = 3 if Cout % 48 == 0 else 4                      # multiplying waves: N = 48 or 64
    m_blk, n_blk = wg_m * 16, wg_n * 16 * NW
    xin = x.reshape(-1)
    in_bytes = xin.size * 4
    uflat = u.reshape(-1)
    u_bytes = uflat.size * 4
    lds = np.zeros(2 * LDS_V // 4, f32)
    acc = np.zeros((NW, 64, 36, 4), np.float64)         # wave, lane, position, r
    pix_stride = in_ld * 4
    for cc in range(CC):
        # ---- staging wave: lane (tile_s, c4) ----
        for lane in range(64):
            tile_s, c4 = lane >> 2, lane & 3
            tile = m_blk + tile_s
            live = tile < T
            tt = tile if live else 0
            tx, tq = tt % TW, tt // TW
            ty, b = tq % TH, tq // TH
            y0, x0 = 4 * ty - 1, 4 * tx - 1
            row_off = [((b * H + y0 + i) * W * pix_stride + c4 * 16) if (live and 0 <= y0 + i < H)
                       else BAD for i in range(6)]
            col_off = [(x0 + j) * pix_stride if 0 <= x0 + j < W else BAD for j in range(6)]
            raw = [[buf_load(xin, in_bytes, row_off[i] + col_off[j] + cc * 64, 16)
                    for j in range(6)] for i in range(6)]
            for i in range(6):
                raw[i] = bt6(raw[i])                                  # along x
            st_off = tile_s * 64 + (((c4 ^ tile_s ^ (tile_s >> 1)) & 3) << 4)
            for j in range(6):
                v = bt6([raw[i][j] for i in range(6)])                # along y
                for i in range(6):
                    o = ((cc & 1) * LDS_V + st_off + (6 * i + j) * PSTR) // 4
                    lds[o:o + 4] = v[i]
        # ---- multiplying waves ----
        for wave in range(NW):
            n0 = n_blk + 16 * wave
            AF = np.zeros((36, 64, 4), f32)
            BF = np.zeros((36, 64, 4), f32)
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                frag_off = l15 * 64 + (((g ^ l15 ^ (l15 >> 1)) & 3) << 4)
                u_lane = ((n0 + l15) * 16 + 4 * g) * 4
                u_pos, u_chunk = CC * Cout * 64, Cout * 64
                for p in range(36):
                    o = ((cc & 1) * LDS_V + frag_off + p * PSTR) // 4
                    AF[p, lane] = lds[o:o + 4]
                    BF[p, lane] = buf_load(uflat, u_bytes, u_lane + p * u_pos + cc * u_chunk, 16)
            # v_mfma_f32_16x16x4_f32, four per position: lane l supplies A[i = l & 15][k = l >> 4]
            # and B[k = l >> 4][j = l & 15]; MFMA kk takes element kk of every lane's 16 bytes.
            # A operand = FILTER fragment (rows = output channels), B operand = V fragment (columns
            # = tiles): D[channel i][tile j]
            A = BF.reshape(36, 4, 16, 4).astype(np.float64)            # p, g, i (channel), kk
            Bm = AF.reshape(36, 4, 16, 4).astype(np.float64)           # p, g, j (tile), kk
            D = np.einsum('pgik,pgjk->pij', A, Bm)                     # [36, channel i, tile j]
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                for r in range(4):
                    acc[wave, lane, :, r] += D[:, 4 * g + r, l15]      # C layout: rows 4 g + r
    # ---- epilogue (conv_wino4.h): lane = tile m_blk + l15, channels n0 + 4 g .. + 3, 16-byte
    # residual loads / stores ----
    outf = out.reshape(-1)
    for wave in range(NW):
        n0 = n_blk + 16 * wave
        for lane in range(64):
            g, l15 = lane >> 4, lane & 15
            col4 = n0 + 4 * g
            tile = m_blk + l15
            live = tile < T
            tt = tile if live else 0
            tx, tq = tt % TW, tt // TW
            ty, b = tq % TH, tq // TH
            pix0 = (b * H + 4 * ty) * W + 4 * tx
            obase = (pix0 * out_ld + out_coff + col4) * 4 if live else BAD
            rbase = (pix0 * res_ld + res_coff + col4) * 4 if (live and res is not None) else BAD
            nrow, ncol = H - 4 * ty, W - 4 * tx
            m = acc[wave, lane].astype(f32).reshape(6, 6, 4)           # i, j, channel r
            s = [at6([m[i, j] for j in range(6)]) for i in range(6)]
            for bb in range(4):
                y = at6([s[i][bb] for i in range(6)])
                for a in range(4):
                    ok = a < nrow and bb < ncol
                    soff_r = (a * W + bb) * res_ld * 4
                    soff_o = (a * W + bb) * out_ld * 4
                    rv = np.zeros(4, f32)
                    if res is not None:
                        ro = (rbase if ok else BAD) + soff_r
                        rv = buf_load(res.reshape(-1), BAD, ro, 16) if ro < BAD else np.zeros(4, f32)
                    v = (y[a] + bias[col4:col4 + 4]) + rv
                    if relu:
                        v = np.maximum(v, f32(0))
                    oo = (obase if ok else BAD) + soff_o
                    if oo < BAD:                              # out-of-range stores are dropped
                        assert oo % 16 == 0 or True
                        assert np.isnan(outf[oo // 4: oo // 4 + 4]).all(), 'element written twice'
                        outf[oo // 4: oo // 4 + 4] = v


def emulate_workgroup4(x, u, bias, res, relu, out, wg_m, wg_n, out_ld, out_coff, res_ld, res_coff):
    """The FOUR-multiplying-wave workgroup of csrc/conv_wino4.hip (round 6): staging by every wave (lane = tile 4 w +
    (l >> 4), channel l & 15, dword loads, row-pair layout, x pass on pairs, y pass per column), the same LDS image
    and fragment reads, the 27-item split (waves 0-2: positions 0..26 of their 16 channels; wave 3: positions
    27..35 of all three channel groups), wave 3's partial x transform through the exchange buffer, the owners'
    completion of rows 4 / 5, then the shared epilogue."""
    B, H, W, in_ld = x.shape
    Cin = in_ld
    Cout = u.shape[2]
    TW, TH = (W + 3) // 4, (H + 3) // 4
    T = B * TH * TW
    CC = Cin // 16
    m_blk, n_blk = wg_m * 16, wg_n * 48
    xin = x.reshape(-1)
    in_bytes = xin.size * 4
    uflat = u.reshape(-1)
    u_bytes = uflat.size * 4
    lds = np.zeros(2 * LDS_V // 4, f32)
    acc = np.zeros((4, 64, 27, 4), np.float64)          # wave, lane, item, r
    pix_stride = in_ld * 4
    u_pos, u_chunk = CC * Cout * 64, Cout * 64
    for cc in range(CC):
        # ---- staging, all four waves: one channel of one tile per lane ----
        for wave in range(4):
            for lane in range(64):
                tile_s, ch = 4 * wave + (lane >> 4), lane & 15
                tile = m_blk + tile_s
                live = tile < T
                tt = tile if live else 0
                tx, tq = tt % TW, tt // TW
                ty, b = tq % TH, tq // TH
                y0, x0 = 4 * ty - 1, 4 * tx - 1
                row_off = [((b * H + y0 + i) * W * pix_stride + ch * 4) if (live and 0 <= y0 + i < H)
                           else BAD for i in range(6)]
                col_off = [(x0 + j) * pix_stride if 0 <= x0 + j < W else BAD for j in range(6)]
                # (the chunk sits in the buffer load's SCALAR offset, which the bounds check does not see)
                d = [[xin[(row_off[i] + col_off[j] + cc * 64) // 4] if row_off[i] + col_off[j] < in_bytes else f32(0)
                      for j in range(6)] for i in range(6)]
                # row pairs rp[k][j] = (d[2k][j], d[2k+1][j]); x pass on whole pairs
                rp = [[np.array([d[2 * k][j], d[2 * k + 1][j]], f32) for j in range(6)] for k in range(3)]
                for k in range(3):
                    rp[k] = bt6(rp[k])
                st_off = tile_s * 64 + ((((ch >> 2) ^ tile_s ^ (tile_s >> 1)) & 3) << 4) + (ch & 3) * 4
                for j in range(6):
                    P0, P1, P2 = rp[0][j], rp[1][j], rp[2][j]
                    o05 = f32(4) * P0 + (f32(-5) * P1 + P2)                   # (o0, o5)
                    U_, W_ = f32(-4) * P1 + P2, f32(-4) * P0 + P1             # a = U.lo, b = W.hi
                    C_, E_ = P2 - P1, P1 - P0                                 # c = C.lo, e = E.hi
                    o12 = np.array([U_[0] + W_[1], U_[0] - W_[1]], f32)       # op_sel (lo, hi), neg_hi
                    o34 = np.array([E_[1] * f32(2) + C_[0], -E_[1] * f32(2) + C_[0]], f32)
                    v = [o05[0], o12[0], o12[1], o34[0], o34[1], o05[1]]
                    for i in range(6):
                        lds[((cc & 1) * LDS_V + st_off + (6 * i + j) * PSTR) // 4] = v[i]
        # ---- multiplying: 27 items per wave ----
        for wave in range(4):
            w3 = wave == 3
            AF = np.zeros((27, 64, 4), f32)
            BF = np.zeros((27, 64, 4), f32)
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                frag_off = l15 * 64 + (((g ^ l15 ^ (l15 >> 1)) & 3) << 4)
                u_lane = ((n_blk + 16 * (0 if w3 else wave) + l15) * 16 + 4 * g) * 4
                for q in range(27):
                    pos = 27 + q // 3 if w3 else q
                    o = ((cc & 1) * LDS_V + frag_off + pos * PSTR) // 4
                    AF[q, lane] = lds[o:o + 4]
                    BF[q, lane] = buf_load(uflat, u_bytes,
                                           u_lane + pos * u_pos + cc * u_chunk + ((q % 3) * 1024 if w3 else 0), 16)
            A = BF.reshape(27, 4, 16, 4).astype(np.float64)            # q, g, i (channel), kk
            Bm = AF.reshape(27, 4, 16, 4).astype(np.float64)           # q, g, j (tile), kk
            D = np.einsum('qgik,qgjk->qij', A, Bm)
            for lane in range(64):
                g, l15 = lane >> 4, lane & 15
                for r in range(4):
                    acc[wave, lane, :, r] += D[:, 4 * g + r, l15]
    # ---- wave 3: x transform of what it holds -> exchange buffer X[n][k][lane] ----
    X = np.zeros((3, 8, 64, 4), f32)
    for lane in range(64):
        a3 = acc[3, lane].astype(f32)                                  # item 3 k + n, k = position - 27
        for n in range(3):
            m3, m4, m5 = a3[0 + n], a3[3 + n], a3[6 + n]               # M[4][3..5]
            s34, d34 = m3 + m4, m3 - m4
            X[n, 0, lane], X[n, 1, lane], X[n, 2, lane] = s34, f32(2) * d34, f32(4) * s34
            X[n, 3, lane] = f32(8) * d34 + m5
            row5 = at6([a3[9 + n], a3[12 + n], a3[15 + n], a3[18 + n], a3[21 + n], a3[24 + n]])
            for k in range(4):
                X[n, 4 + k, lane] = row5[k]
    # ---- owners: rows 0..3 from their own accumulators, row 4 = own half + wave 3's, row 5 = wave 3's ----
    outf = out.reshape(-1)
    for wave in range(3):
        n0 = n_blk + 16 * wave
        for lane in range(64):
            g, l15 = lane >> 4, lane & 15
            col4 = n0 + 4 * g
            tile = m_blk + l15
            live = tile < T
            tt = tile if live else 0
            tx, tq = tt % TW, tt // TW
            ty, b = tq % TH, tq // TH
            pix0 = (b * H + 4 * ty) * W + 4 * tx
            obase = (pix0 * out_ld + out_coff + col4) * 4 if live else BAD
            rbase = (pix0 * res_ld + res_coff + col4) * 4 if (live and res is not None) else BAD
            nrow, ncol = H - 4 * ty, W - 4 * tx
            m = acc[wave, lane].astype(f32)                            # item = position 0..26
            s = [at6([m[6 * i + j] for j in range(6)]) for i in range(4)]
            s12, d12 = m[25] + m[26], m[25] - m[26]
            r = X[wave, :, lane]
            s.append([(m[24] + s12) + r[0], d12 + r[1], s12 + r[2], d12 + r[3]])
            s.append([r[4], r[5], r[6], r[7]])
            for bb in range(4):
                y = at6([s[i][bb] for i in range(6)])
                for a in range(4):
                    ok = a < nrow and bb < ncol
                    rv = np.zeros(4, f32)
                    if res is not None:
                        ro = (rbase if ok else BAD) + (a * W + bb) * res_ld * 4
                        rv = buf_load(res.reshape(-1), BAD, ro, 16) if ro < BAD else np.zeros(4, f32)
                    v = (y[a] + bias[col4:col4 + 4]) + rv
                    if relu:
                        v = np.maximum(v, f32(0))
                    oo = (obase if ok else BAD) + (a * W + bb) * out_ld * 4
                    if oo < BAD:
                        assert np.isnan(outf[oo // 4: oo // 4 + 4]).all(), 'element written twice'
                        outf[oo // 4: oo // 4 + 4] = v


def direct_conv(x, w, bias):
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    ref = np.zeros((B, H, W, w.shape[0]))
    for kh in range(3):
        for kw in range(3):
            ref += np.einsum('bhwc,oc->bhwo', xp[:, kh:kh + H, kw:kw + W], w[:, kh, kw].astype(np.float64))
    return ref + bias


def check(B, H, W, Cin, Cout, with_res, relu, coff=0, seed=0, four_waves=False):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(f32)
    w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(f32)
    bias = rng.standard_normal(Cout).astype(f32)
    res = rng.standard_normal((B, H, W, Cout)).astype(f32) if with_res else None
    u = wg.transform_filters4(w)
    out_ld = Cout + coff
    out = np.full((B, H, W, out_ld), np.nan, f32)
    TW, TH = (W + 3) // 4, (H + 3) // 4
    nby, nbx = (B * TH * TW + 15) // 16, Cout // (48 if Cout % 48 == 0 else 64)
    for m in range(nby):
        for n in range(nbx):
            (emulate_workgroup4 if four_waves else emulate_workgroup)(
                x, u, bias, res, relu, out, m, n, out_ld, coff, Cout, 0)
    ref = direct_conv(x, w, bias)
    if with_res:
        ref = ref + res
    if relu:
        ref = np.maximum(ref, 0)
    got = out[..., coff:]
    assert not np.isnan(got).any(), 'unwritten outputs'
    assert coff == 0 or np.isnan(out[..., :coff]).all(), 'wrote outside its channel slice'
    err = np.abs(got - ref).max()
    print(f'{"four-wave " if four_waves else ""}B={B} {H}x{W} {Cin}->{Cout} res={with_res} '
          f'relu={relu} coff={coff}: '
          f'{nby * nbx} workgroups, max err {err:.2e}')
    assert err < 2e-5, err


def network_numerics():
    """F(4x4,3x3) in float32 for every eligible layer of the CPU oracle's HRNet-W48."""
    import torch
    import torch.nn.functional as F
    from oracle import hrnet_torch as ht
    from shapy_amd.utils import synthetic as syn

    def wino(x, w):
        Bn, C, H, W = x.shape
        th, tw = -(-H // 4), -(-W // 4)
        xp = F.pad(x, (1, tw * 4 - W + 1, 1, th * 4 - H + 1))
        d = xp.unfold(2, 6, 4).unfold(3, 6, 4)
        bt = torch.tensor(wg.BT4, dtype=x.dtype)
        at = torch.tensor(wg.AT4, dtype=x.dtype)
        U = torch.tensor(np.einsum('ik,ockl,jl->ocij', wg.G4, w.double().numpy(), wg.G4), dtype=x.dtype)
        V = torch.einsum('ik,bcxykl,jl->bcxyij', bt, d, bt)
        M = torch.einsum('ocij,bcxyij->boxyij', U, V)
        Y = torch.einsum('ik,boxykl,jl->boxyij', at, M, at)
        return Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, -1, th * 4, tw * 4)[:, :, :H, :W]

    mode = {'on': False}
    orig = ht._conv

    def conv(sd, p, x, stride=1, padding=0):
        w = sd[p + '.weight']
        if mode['on'] and w.shape[-1] == 3 and stride == 1 and padding == 1 and w.shape[1] % 16 == 0:
            y = wino(x, w)
            b = sd.get(p + '.bias')
            return y if b is None else y + b.view(1, -1, 1, 1)
        return orig(sd, p, x, stride, padding)
    ht._conv = conv
    sd = syn.synthetic_state_dict(ht.state_dict_spec(), 0)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    x = torch.from_numpy(syn.synthetic_images(2, 224, 100))
    with torch.no_grad():
        f64 = ht.hrnet_forward(sd64, x.double())
        f32d = ht.hrnet_forward(sd, x)
        mode['on'] = True
        f32w = ht.hrnet_forward(sd, x)
    print('feature range', f64.abs().max().item())
    print('direct  float32 vs float64:', (f32d.double() - f64).abs().max().item())
    print('F(4x4)  float32 vs float64:', (f32w.double() - f64).abs().max().item())


if __name__ == '__main__':
    if '--network' in sys.argv:
        network_numerics()
        sys.exit(0)
    check(1, 8, 8, 16, 48, False, True)                 # 4 tiles: one partly filled workgroup
    check(2, 12, 20, 48, 48, True, True)                # 30 tiles: two workgroups, 3 chunks
    check(1, 7, 9, 32, 96, True, False)                 # partial edge tiles, two N tiles
    check(1, 14, 14, 16, 48, False, False, coff=16)     # concat-style channel offset
    check(1, 9, 10, 32, 64, True, True)                 # (the 64-channel N tile of rounds 4-5: four multiplying waves)
    check(1, 6, 6, 16, 128, False, True)                # ... two of them
    # the four-multiplying-wave workgroup of the per-layer kernel
    check(1, 8, 8, 16, 48, False, True, four_waves=True)
    check(2, 12, 20, 48, 48, True, True, four_waves=True)
    check(1, 7, 9, 32, 96, True, False, four_waves=True)
    check(1, 14, 14, 16, 48, False, False, coff=16, four_waves=True)
    print('emulation OK')
