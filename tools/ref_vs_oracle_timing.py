"""CPU wall time of the REAL reference modules next to the CPU oracle on the same inputs.

    python tools/ref_vs_oracle_timing.py [--batch 4] [--threads 8] [--repeat 5] [--out file.json]

Runs in the build container only (it imports /root/reference through tests/golden/ref_loader.py;
the reference tree does not exist on the GPU box).  `bench.py`'s `cpu_baseline` has to use the
oracle there (`kind: "port"`); this tool measures, where both can run, how far the oracle's wall
time is from the reference's own (reference modules on CPU, N threads and 1 thread,
warm-up 1, median of >= 5) and that both produce the same numbers.  The result is committed under
profiles/ and cited in DESIGN.md.
"""
import argparse
import json
import os
import os.path as osp
import statistics
import sys
import time

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, 'tests', 'golden'))

import ref_loader                                     # noqa: E402
import __graft_entry__ as ge                          # noqa: E402
from oracle import measure as omeasure                # noqa: E402
from shapy_amd.config import merge_config             # noqa: E402
from shapy_amd.utils import synthetic as syn          # noqa: E402


def median_time(fn, repeat):
    fn()                                              # warm-up
    ts = []
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def cpu_model():
    with open('/proc/cpuinfo') as fh:
        return next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), '')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--repeat', type=int, default=5)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    if not ref_loader.available():
        raise SystemExit('the reference tree is not mounted here')

    ns = ref_loader.load_reference(intersect_fn=omeasure.mesh_to_mesh_forward)
    data = osp.join(ROOT, 'shapy_amd', 'data')
    model_folder = '/tmp/shapy_synth_models'
    syn.write_synthetic_smplx(model_folder, 0)
    cfg = merge_config([osp.join(ROOT, 'configs/b2a_expose_hrnet_demo.yaml')], [
        f'body_model.model_folder={model_folder}',
        'network.smplx.backbone.hrnet.pretrained_path=',
        f'network.smplx.meas_definition_path={data}/measurement_defitions.yaml',
        f'network.smplx.meas_vertices_path={data}/smplx_measurements.yaml'])
    ref = ns.body_heads.BODY_HEAD_REGISTRY['SMPLXRegressor'](
        cfg.body_model, network_cfg=cfg.network.smplx, loss_cfg=cfg.losses.body).eval()
    syn.fill_module_synthetic(ref, 0)
    state = ge.oracle_state(0)
    x_np = syn.synthetic_images(args.batch, args.size, 100)
    x = torch.from_numpy(x_np)

    def run_ref():
        with torch.no_grad():
            return ref(x, None)

    def run_oracle():
        return ge.oracle_forward(x_np, state=state)

    rows = {}
    for threads in sorted({args.threads, 1}, reverse=True):
        torch.set_num_threads(threads)
        t_ref, o_ref = median_time(run_ref, args.repeat)
        t_ora, o_ora = median_time(run_oracle, args.repeat)
        rows[f'threads_{threads}'] = {
            'reference_s': t_ref, 'oracle_s': t_ora,
            'reference_images_per_s': args.batch / t_ref, 'oracle_images_per_s': args.batch / t_ora,
            'oracle_over_reference_time': t_ora / t_ref}
    st = o_ref['stage_02']
    err = {
        'features': float(np.abs(o_ref['features'].numpy() - o_ora['features']).max()),
        'betas': float(np.abs(st['betas'].numpy() - o_ora['stages'][-1]['betas']).max()),
        'vertices': float(np.abs(st['vertices'].numpy() - o_ora['stages'][-1]['vertices']).max()),
    }
    if 'measurements' in o_ref:
        m = o_ref['measurements']
        for k in ('mass', 'height', 'chest', 'waist', 'hips'):
            v = m[k]['tensor'] if isinstance(m[k], dict) else m[k]
            err[k] = float(np.abs(np.asarray(v.detach().numpy()).reshape(-1) -
                                  o_ora['measurements'][k].reshape(-1)).max())
    res = {'what': 'wall time of one forward of the hot path on CPU: the reference\'s own '
                   'SMPLXRegressor (regressor/human_shape, imported from /root/reference; its '
                   'CUDA-only intersection op replaced by the C oracle) vs the oracle '
                   '(__graft_entry__.oracle_forward), same seeded weights and images; warm-up 1, '
                   f'median of {args.repeat}',
           'batch': args.batch, 'size': args.size, 'cpu_model': cpu_model(),
           'host_logical_cpus': os.cpu_count(), 'timings': rows,
           'max_abs_difference_oracle_vs_reference': err}
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(txt + '\n')


if __name__ == '__main__':
    main()
