"""Phase timing of bvh_build_kernel (tuning only): variant library with -DSHAPY_BVH_TIMING, wall_clock64
stamps of workgroup 0: bounds / keys / block sorts / cross-block merge / radix tree / leaf boxes / refit."""
import ctypes
import os
import os.path as osp
import subprocess
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = '/tmp/libshapy_bvh_timing.so'


def main():
    env = dict(os.environ, SHAPY_HIPCC_FLAGS='-DSHAPY_BVH_TIMING', SHAPY_HIP_LIB=VARIANT)
    subprocess.check_call([sys.executable, '-m', 'shapy_amd.build'], cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
    os.environ['SHAPY_HIP_LIB'] = VARIANT
    import torch
    import bench
    import mesh_mesh_intersect_cuda as mmi
    from shapy_amd import _lib
    lib = _lib.load()
    lib.shapy_debug_bvh_times.restype = ctypes.c_int
    lib.shapy_debug_bvh_times.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    for n in (1, 256, 1000):
        faces_np, v_np = bench.config4_meshes(n, 0)
        tris = torch.from_numpy(v_np).cuda()[:, torch.from_numpy(faces_np).cuda().long()].contiguous()
        query = torch.roll(tris, -1, 0)[:, 3000:3700].contiguous()
        for _ in range(3):
            mmi.mesh_to_mesh_forward(query, tris, max_collisions=32)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 16)()
        assert lib.shapy_debug_bvh_times(buf) == 0
        names = ['bounds', 'keys', 'block sorts (LDS)', 'cross-block merge', 'radix tree', 'leaf boxes', 'refit']
        print(f'{n} meshes, workgroup 0: ' + ', '.join(f'{nm} {(buf[i + 1] - buf[i]) / 100.0:.1f} us' for i, nm in enumerate(names)),
              f'| total {(buf[7] - buf[0]) / 100.0:.1f} us')


if __name__ == '__main__':
    main()
