"""Phase timing of one workgroup of the persistent grouped F(4x4) kernel (tuning only): builds a
variant library with -DSHAPY_W4G_TIMING and prints the wall_clock64 stamps of a mid-grid workgroup
(one multiplying wave, the staging wave) for a few groups at B = 64.

    python tools/wino4g_timing.py       # on a GPU box
"""
import ctypes
import os
import os.path as osp
import subprocess
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = '/tmp/libshapy_w4g_timing.so'
MULT = {1: 'task start', 2: 'at barrier', 3: 'barrier passed', 4: 'epilogue start', 5: 'epilogue end'}
STAGE = {0: 'start', 1: 'task top', 3: 'chunk staged', 4: 'barrier passed'}


def main():
    env = dict(os.environ, SHAPY_HIPCC_FLAGS='-DSHAPY_W4G_TIMING', SHAPY_HIP_LIB=VARIANT)
    subprocess.check_call([sys.executable, '-m', 'shapy_amd.build'], cwd=ROOT, env=env,
                          stdout=subprocess.DEVNULL)
    os.environ['SHAPY_HIP_LIB'] = VARIANT
    import torch
    from shapy_amd import _lib
    sys.path.insert(0, osp.join(ROOT, 'tools'))
    import wino4g_check as chk
    lib = _lib.load()
    lib.shapy_debug_w4g_times.restype = ctypes.c_int
    lib.shapy_debug_w4g_times.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B = 64
    groups = {'48 alone': [(B, 56, 56, 48, 48, True, True)],
              'stage3': [(B, 56, 56, 48, 48, True, True), (B, 28, 28, 96, 96, True, True),
                         (B, 14, 14, 192, 192, True, True)]}
    for name, shapes in groups.items():
        g = torch.Generator().manual_seed(1)
        descs, keep = [], []
        for s in shapes:
            d, k = chk.make_desc(*s, g=g)
            descs.append(d); keep.append(k)
        arr = (_lib.ShapyConv * len(descs))(*descs)
        for _ in range(3):
            assert lib.shapy_conv2d_group(arr, len(descs), stream) == 0
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 256)()
        n = (ctypes.c_int * 2)()
        assert lib.shapy_debug_w4g_times(buf, n) == 0
        print(f'== {name}: stamps mult {n[0]} staging {n[1]}')
        t0 = min(buf[i] & ((1 << 56) - 1) for r in range(2) for i in range(r * 128, r * 128 + min(n[r], 128)))
        ev = []
        for r, names in ((0, MULT), (1, STAGE)):
            for i in range(min(n[r], 128)):
                v = buf[r * 128 + i]
                ev.append((((v & ((1 << 56) - 1)) - t0) / 100.0, 'mult ' if r == 0 else 'stage', names.get(v >> 56, '?')))
        last = {}
        for t, who, what in sorted(ev):
            print(f'   {t:8.2f} us  {who}  {what:16s} (+{t - last.get(who, 0.0):.2f})')
            last[who] = t


if __name__ == '__main__':
    main()
