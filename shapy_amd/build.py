"""Builds libshapy_hip.so (gfx950) in-tree with hipcc -- no torch, no cmake.

    python -m shapy_amd.build [--force]

Each .hip file is compiled to an object (parallel), then linked.  hipcc cross-compiles for
gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import os
import os.path as osp
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = osp.dirname(osp.abspath(__file__))
CSRC = osp.join(HERE, 'csrc')
OUT = osp.join(CSRC, 'libshapy_hip.so')
ARCH = 'gfx950'

SOURCES = {
    'conv_igemm.hip': [],
    'conv_x6.hip': [],
    'conv_wino.hip': [],
    'conv_wino4.hip': [],
    'conv_wino4g.hip': [],
    'hrnet_ops.hip': [],
    'body.hip': [],
    # bit-identical float32 decisions with the CPU oracle: no FMA contraction here
    'measure.hip': ['-ffp-contract=off'],
    'bvh.hip': ['-ffp-contract=off'],
    'preprocess.hip': ['-ffp-contract=off'],
    'metrics.hip': [],
    'capi.hip': [],
}
COMMON = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function']
# tuning experiments: SHAPY_HIPCC_FLAGS='-DSHAPY_MFMA_PRIO=2' SHAPY_HIP_LIB=/path/variant.so
COMMON += os.environ.get('SHAPY_HIPCC_FLAGS', '').split()
OUT = os.environ.get('SHAPY_HIP_LIB', OUT)


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (osp.isabs(c) and osp.exists(c) or not osp.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _newest_dep():
    deps = [osp.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h', '.map'))]
    deps.append(osp.join(osp.dirname(HERE), 'include', 'shapy_hip.h'))
    return max(osp.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    if not force and osp.exists(OUT) and osp.getmtime(OUT) >= _newest_dep():
        return OUT
    cc = hipcc()
    objdir = osp.join(CSRC, 'build' if OUT.endswith('libshapy_hip.so') else 'build_' + osp.basename(OUT))
    os.makedirs(objdir, exist_ok=True)

    def compile_one(item):
        src, extra = item
        obj = osp.join(objdir, src.replace('.hip', '.o'))
        cmd = [cc] + COMMON + extra + ['-c', osp.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES.items()))
    cmd = [cc, '-shared', '-fPIC', f'--offload-arch={ARCH}',
           '-Wl,--version-script=' + osp.join(CSRC, 'exports.map'), '-o', OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
