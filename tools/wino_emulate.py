"""Thread-by-thread NumPy emulation of one conv_wino_kernel<3, TM> workgroup (csrc/conv_wino.hip):
the staging thread map, the DPP quad exchange, the XOR-swizzled LDS image, the MFMA fragment
addressing, the accumulator parking and the output transform -- everything except the hardware
semantics of v_mfma / DPP themselves.  Compares the emulated workgroup output with
shapy_amd.utils.winograd.conv_reference.  CPU only:  python tools/wino_emulate.py
"""
import os.path as osp
import sys

import numpy as np

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from shapy_amd.utils import winograd as wg      # noqa: E402

f32 = np.float32


def emulate_workgroup(x, u, bias, wg_m, wg_n, TM=1, NN=3):
    N, NCP = 16 * NN, 16 * NN + 4
    B, H, W, Cin = x.shape
    Cout = u.shape[2]
    TW, TH = (W + 1) // 2, (H + 1) // 2
    T = B * TH * TW
    CC = Cin // 16
    MT, PSTR = 16 * TM, 16 * TM * 64
    m_blk, n_blk = wg_m * MT, wg_n * N
    lds = np.zeros(16 * PSTR // 4, f32)                        # one V buffer, as floats
    acc = np.zeros((4, 64, 4, TM, NN, 4), np.float64)          # wave, lane, pp, m, n, rg
    uflat = u.reshape(-1)
    for cc in range(CC):
        # ---- staging: every thread computes T[r][j][e], then the quad exchange ----
        for sg in range(TM):
            tr = np.zeros((256, 4, 4), f32)
            meta = []
            for t in range(256):
                r, c4, tile_s = t & 3, (t >> 3) & 3, ((t >> 5) << 1) | ((t >> 2) & 1)
                tile = m_blk + 16 * sg + tile_s
                d = np.zeros((4, 4), f32)
                if tile < T:
                    tx, tq = tile % TW, tile // TW
                    ty, b = tq % TH, tq // TH
                    y, x0 = 2 * ty - 1 + r, 2 * tx - 1
                    if 0 <= y < H:
                        for k in range(4):
                            if 0 <= x0 + k < W:
                                d[k] = x[b, y, x0 + k, cc * 16 + c4 * 4: cc * 16 + c4 * 4 + 4]
                tr[t, 0] = d[0] - d[2]
                tr[t, 1] = d[1] + d[2]
                tr[t, 2] = d[2] - d[1]
                tr[t, 3] = d[1] - d[3]
                meta.append((r, c4, tile_s))
            for t in range(256):
                r, c4, tile_s = meta[t]
                partner = (t & ~3) | (2, 2, 1, 1)[r]                  # quad_perm:[2,2,1,1]
                so = -1.0 if r == 3 else 1.0
                sp = 1.0 if r in (1, 3) else -1.0
                fsw = (tile_s ^ (tile_s >> 1)) & 3
                st_off = tile_s * 64 + (((c4 ^ fsw ^ r) & 3) << 4) + 4 * r * PSTR
                for j in range(4):
                    v = (f32(sp) * tr[partner, j] + f32(so) * tr[t, j]).astype(f32)
                    o = (st_off + sg * 1024 + j * PSTR) // 4
                    lds[o:o + 4] = v
        # ---- MFMA role ----
        for wave in range(4):
            for pp in range(4):
                p = 4 * wave + pp
                for m in range(TM):
                    af = np.zeros((64, 4), f32)
                    for lane in range(64):
                        kq, l15 = lane >> 4, lane & 15
                        frag_off = l15 * 64 + (((kq ^ ((l15 ^ (l15 >> 1)) & 3) ^ wave) & 3) << 4)
                        o = (p * PSTR + m * 1024 + frag_off) // 4
                        af[lane] = lds[o:o + 4]
                    for n in range(NN):
                        bf = np.zeros((64, 4), f32)
                        for lane in range(64):
                            kq, l15 = lane >> 4, lane & 15
                            u_lane = ((n_blk + l15) * 16 + 4 * kq) * 4
                            base = u_lane + p * (CC * Cout * 64) + cc * (Cout * 64) + n * 1024
                            bf[lane] = uflat[base // 4: base // 4 + 4]
                        # D[i][j] += sum_k A[i][k] B[k][j]; A lane: i = l & 15, k = l >> 4
                        for kk in range(4):
                            A = af[:, kk].reshape(4, 16).T          # [i, k]
                            Bm = bf[:, kk].reshape(4, 16)           # [k, j]
                            D = A.astype(np.float64) @ Bm.astype(np.float64)   # [tile, cout]
                            for lane in range(64):
                                kq, l15 = lane >> 4, lane & 15
                                for rg in range(4):
                                    acc[wave, lane, pp, m, n, rg] += D[4 * kq + rg, l15]
    out = {}
    for mt in range(TM):
        # ---- park tile group mt ----
        M = np.zeros(16 * 16 * NCP, f32)
        for wave in range(4):
            for lane in range(64):
                kq, l15 = lane >> 4, lane & 15
                for pp in range(4):
                    for n in range(NN):
                        for rg in range(4):
                            M[((4 * wave + pp) * 16 + 4 * kq + rg) * NCP + n * 16 + l15] = \
                                acc[wave, lane, pp, mt, n, rg]
        for it in range(16 * (N // 4)):
            c4o, tl = it % (N // 4), it // (N // 4)
            tile = m_blk + 16 * mt + tl
            if tile >= T:
                continue
            col = n_blk + c4o * 4
            tx, tq = tile % TW, tile // TW
            ty, b = tq % TH, tq // TH
            tt = np.zeros((4, 2, 4), f32)
            for i in range(4):
                m = [M[((i * 4 + j) * 16 + tl) * NCP + c4o * 4:][:4] for j in range(4)]
                tt[i, 0] = (m[0] + m[1]) + m[2]
                tt[i, 1] = (m[1] - m[2]) - m[3]
            bb_ = bias[col:col + 4]
            for bb in range(2):
                y0 = ((tt[0, bb] + tt[1, bb]) + tt[2, bb]) + bb_
                y1 = ((tt[1, bb] - tt[2, bb]) - tt[3, bb]) + bb_
                for a, yv in ((0, y0), (1, y1)):
                    if 2 * ty + a < H and 2 * tx + bb < W:
                        out[(b, 2 * ty + a, 2 * tx + bb, col)] = yv
    return out


def main():
    r = np.random.default_rng(1)
    B, H, W, Cin, Cout = 1, 7, 10, 32, 192
    x = r.standard_normal((B, H, W, Cin)).astype(f32)
    w = (r.standard_normal((Cout, 3, 3, Cin)) * 0.1).astype(f32)
    bias = r.standard_normal(Cout).astype(f32)
    u = wg.transform_filters(w)
    ref = wg.conv_reference(x, u, bias)
    T = B * ((H + 1) // 2) * ((W + 1) // 2)
    for TM, NN in ((1, 3), (2, 3), (1, 4)):
        if Cout % (16 * NN):
            continue
        worst, n = 0.0, 0
        for wm in range((T + 16 * TM - 1) // (16 * TM)):
            for wn in range(Cout // (16 * NN)):
                for (b, y, xx, col), v in emulate_workgroup(x, u, bias, wm, wn, TM, NN).items():
                    worst = max(worst, float(np.abs(v - ref[b, y, xx, col:col + 4]).max()))
                    n += 4
        print(f'TM={TM} NN={NN}: emulated outputs', n, 'of', ref.size, 'max |emulation - reference| =', worst)
        assert n == ref.size and worst < 1e-5
    print('OK')


if __name__ == '__main__':
    main()
