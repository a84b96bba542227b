"""Virtual measurements from betas -- drop-in for the reference's
measurements/virtual_measurements.py (same CLI; rendering is out of scope):

    python measurements/virtual_measurements.py --input-folder ../samples/shapy_fit_for_virtual_measurements \
        --output-folder out --smpl_model_path ../data/body_models

Reads every ``*.npz`` with a ``betas`` entry, evaluates the SMPL-X shape blend shapes and the
fused measurement kernels, prints the values (virtual_measurements.py:57-91) and writes them to
``<output-folder>/<name>_measurements.json``.
"""
import argparse
import json
import os
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)

from shapy_amd.measurements import BodyMeasurements          # noqa: E402
from shapy_amd.models.body_models import SMPLX               # noqa: E402


@torch.no_grad()
def main(demo_input_folder, demo_output_folder, meas_definition_path, meas_vertices_path,
         smpl_model_path, gender='neutral', num_betas=10):
    if not torch.cuda.is_available():
        print('No GPU is available!')
        sys.exit(3)                                    # virtual_measurements.py:33-36
    device = torch.device('cuda')
    os.makedirs(demo_output_folder, exist_ok=True)
    bm = BodyMeasurements({'meas_definition_path': meas_definition_path,
                           'meas_vertices_path': meas_vertices_path}).to(device)
    smpl = SMPLX(osp.join(smpl_model_path, 'smplx'), gender=gender, betas={'num': num_betas},
                 ext='npz').to(device)
    faces = smpl.faces_tensor.to(torch.int32).contiguous()
    results = {}
    for npz_file in sorted(x for x in os.listdir(demo_input_folder) if x.endswith('npz')):
        print(f'Processing: {npz_file}')
        betas = np.load(osp.join(demo_input_folder, npz_file), allow_pickle=True)['betas']
        betas = torch.from_numpy(np.asarray(betas, np.float32)).to(device).reshape(1, -1)
        v_shaped = smpl.forward_shape(betas)['v_shaped']
        vals = bm.forward_vertices(v_shaped, faces)[0].cpu().numpy()
        bm.check_overflow()
        meas = dict(zip(bm.NAMES, [float(v) for v in vals]))
        print('    Virtual measurements: ' + ''.join(
            f'    {k}: {v:.2f} {"kg" if k == "mass" else "m"}' for k, v in meas.items()))
        with open(osp.join(demo_output_folder, npz_file.replace('.npz', '_measurements.json')), 'w') as f:
            json.dump(meas, f)
        results[npz_file] = meas
    return results


if __name__ == '__main__':
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                description='Virtual measurements (MI355X)')
    p.add_argument('--output-folder', dest='output_folder', default='demo_output', type=str)
    p.add_argument('--input-folder', dest='input_folder', default='demo_input', type=str)
    p.add_argument('--meas_definition_path', default=osp.join(
        ROOT, 'shapy_amd', 'data', 'measurement_defitions.yaml'), type=str)
    p.add_argument('--meas_vertices_path', default=osp.join(
        ROOT, 'shapy_amd', 'data', 'smplx_measurements.yaml'), type=str)
    p.add_argument('--smpl_model_path', default='../data/body_models', type=str)
    p.add_argument('--num_betas', default=10, type=int)
    p.add_argument('--gender', default='neutral', type=str)
    a = p.parse_args()
    main(a.input_folder, a.output_folder, a.meas_definition_path, a.meas_vertices_path,
         a.smpl_model_path, a.gender, a.num_betas)
