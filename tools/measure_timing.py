"""Phase timing of the measurement kernels (tuning only).  Builds a variant library with
-DSHAPY_MEASURE_TIMING (wall_clock64 stamps of workgroup (0, mesh 7), 100 MHz ticks) and prints
the phases of measure_scan2_kernel / measure_hull2_kernel for a few batch sizes.

    python tools/measure_timing.py            # on a GPU box
"""
import ctypes
import os
import os.path as osp
import subprocess
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = '/tmp/libshapy_timing.so'


def main():
    env = dict(os.environ, SHAPY_HIPCC_FLAGS='-DSHAPY_MEASURE_TIMING', SHAPY_HIP_LIB=VARIANT)
    subprocess.check_call([sys.executable, '-m', 'shapy_amd.build'], cwd=ROOT, env=env,
                          stdout=subprocess.DEVNULL)
    os.environ['SHAPY_HIP_LIB'] = VARIANT
    import torch
    import bench
    from shapy_amd import _lib
    from shapy_amd.measurements import BodyMeasurements
    lib = _lib.load()
    lib.shapy_debug_measure_times.restype = ctypes.c_int
    lib.shapy_debug_measure_times.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    data = osp.join(ROOT, 'shapy_amd', 'data')
    bm = BodyMeasurements({'meas_definition_path': f'{data}/measurement_defitions.yaml',
                           'meas_vertices_path': f'{data}/smplx_measurements.yaml'}).cuda()
    names = {0: 'scan start', 1: 'staged', 2: 'scan loop done', 3: 'candidates done',
             4: 'volume written', 8: 'hull start', 9: 'gathered', 10: 'sorted', 11: 'compacted',
             12: 'chains done', 13: 'perimeter', 14: 'written'}
    for n in (64, 1000):
        faces, v = bench.config4_meshes(n)
        vt, ft = torch.from_numpy(v).cuda(), torch.from_numpy(faces).cuda()
        for _ in range(3):
            bm.forward_vertices(vt, ft)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 32)()
        assert lib.shapy_debug_measure_times(buf) == 0
        t = {k: buf[k] for k in names}
        print(f'--- {n} meshes (mesh 7, slice 0 / plane 0), microseconds since kernel-local start')
        for grp in ((0, 1, 2, 3, 4), (8, 9, 10, 11, 12, 13, 14)):
            t0 = t[grp[0]]
            print('   ' + '  '.join(f'{names[k]}: {(t[k] - t0) / 100.0:.1f}' for k in grp))


if __name__ == '__main__':
    main()
