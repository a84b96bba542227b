"""HBW evaluation -- drop-in for the reference's regressor/hbw_evaluation/evaluate_hbw.py
(same CLI and printed report, :61-187):

    python hbw_evaluation/evaluate_hbw.py --input-npz-file pred.npz --hbw-folder datasets/HBW \
        --model-type smplx --point-reg-gt HD_SMPLX_from_SMPL.pkl --point-reg-fit HD_SMPLX_from_SMPL.pkl

``pred.npz`` holds ``image_name`` ("split/subject_xx/.../img") and ``v_shaped`` [N,V,3]; the
ground truth is ``<hbw>/smplx/<split>/<subject>.npy``.  For every sample: V2V error after
translation alignment (SMPL-X fits only), P2P-20k error (two sparse point regressors,
float64) and |gt - fit| of height / chest / waist / hips / mass.

Where the reference loops over the samples on the CPU (numpy point errors, one
BodyMeasurements call + scipy hull per mesh and side), this runs the whole set in batches on the
GPU: ``shapy_aligned_point_error_f32``, ``shapy_p2p_error_f64`` and ``shapy_body_measure_f32``
(``shapy_amd/utils/metrics.py``, ``shapy_amd/measurements``).  The reference builds the body
models with the pip package ``smplx`` only to read their face tables; here the faces come from
the same model files (``<body-model-folder>/<type>/<TYPE>_NEUTRAL.npz``).
"""
import argparse
import os.path as osp
import pickle
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)

from shapy_amd.measurements import BodyMeasurements          # noqa: E402
from shapy_amd.utils import metrics                          # noqa: E402

PARENT_FOLDER = osp.dirname(osp.abspath(__file__))
DEFAULT_HBW_FOLDER = osp.join(PARENT_FOLDER, '..', 'datasets', 'HBW')
_EVAL = osp.join(PARENT_FOLDER, '..', 'data', 'utility_files', 'evaluation', 'eval_point_set')
DEFAULT_POINT_REG_SMPLX = osp.join(_EVAL, 'HD_SMPLX_from_SMPL.pkl')
DEFAULT_POINT_REG_SMPL = osp.join(_EVAL, 'HD_SMPL_sparse.pkl')
DEFAULT_BODY_MODEL_FOLDER = osp.join(PARENT_FOLDER, '..', 'data', 'body_models')
DEFAULT_BODY_MEASUREMENT_FOLDER = osp.join(ROOT, 'shapy_amd', 'data')
MEAS = ('height', 'chest', 'waist', 'hips', 'mass')


def load_faces(body_model_folder, model_type):
    """Face table of a body model file (the only thing evaluate_hbw.py:101-117 uses the
    ``smplx`` package for)."""
    name = f'{model_type.upper()}_NEUTRAL'
    path = osp.join(body_model_folder, model_type, name + '.npz')
    if osp.exists(path):
        return np.asarray(np.load(path, allow_pickle=True)['f'], np.int64)
    pkl = osp.join(body_model_folder, model_type, name + '.pkl')
    if osp.exists(pkl):
        raise NotImplementedError(
            f'{pkl}: chumpy pickles are not read here; convert the model to npz '
            '(the smplx package ships tools/clean_ch.py) or pass --body-model-folder with npz files')
    raise FileNotFoundError(path)


def ground_truth(labels, hbw_folder):
    """evaluate_hbw.py:134-138: one v_shaped per subject."""
    cache, out = {}, []
    for label in labels:
        split, subject = str(label).split('/')[:2]
        key = (split, subject.split('_')[0])
        if key not in cache:
            cache[key] = np.load(osp.join(hbw_folder, 'smplx', key[0], key[1] + '.npy')).astype(np.float32)
        out.append(cache[key])
    return np.stack(out)


@torch.no_grad()
def evaluate(labels, fits, hbw_folder, model_type='smplx', point_reg_gt=DEFAULT_POINT_REG_SMPLX,
             point_reg_fit=DEFAULT_POINT_REG_SMPLX,
             body_measurement_folder=DEFAULT_BODY_MEASUREMENT_FOLDER,
             body_model_folder=DEFAULT_BODY_MODEL_FOLDER, batch_size=256, device='cuda'):
    """-> dict of per-sample arrays: 'v2v' (SMPL-X only), 'p2p', and one |gt - fit| per
    measurement (metres / kg)."""
    if not torch.cuda.is_available():
        raise RuntimeError('evaluate_hbw runs on the GPU (no CPU fallback)')
    with open(point_reg_gt, 'rb') as f:
        reg_gt = pickle.load(f)
    with open(point_reg_fit, 'rb') as f:
        reg_fit = pickle.load(f)
    meas_def = osp.join(body_measurement_folder, 'measurement_defitions.yaml')
    fit_yaml = (f'{model_type}_measurement_vertices.yaml' if model_type == 'smpl'
                else f'{model_type}_measurements.yaml')
    bm_gt = BodyMeasurements({'meas_definition_path': meas_def, 'meas_vertices_path':
                              osp.join(body_measurement_folder, 'smplx_measurements.yaml')}).to(device)
    bm_fit = BodyMeasurements({'meas_definition_path': meas_def, 'meas_vertices_path':
                               osp.join(body_measurement_folder, fit_yaml)}).to(device)
    f_gt = torch.from_numpy(load_faces(body_model_folder, 'smplx').astype(np.int32)).to(device)
    f_fit = f_gt if model_type == 'smplx' else torch.from_numpy(
        load_faces(body_model_folder, model_type).astype(np.int32)).to(device)
    p2p = metrics.v2vhdError(input_point_regressor=reg_gt, target_point_regressor=reg_fit,
                             align=True).to(device)
    v2v = metrics.PointError(metrics.build_alignment('translation'))
    gts = ground_truth(labels, hbw_folder)
    fits = np.asarray(fits, np.float32)
    res = {k: [] for k in ('v2v', 'p2p') + MEAS}
    for s in range(0, len(fits), batch_size):
        g = torch.from_numpy(gts[s:s + batch_size]).to(device)
        x = torch.from_numpy(fits[s:s + batch_size]).to(device)
        if model_type == 'smplx':
            res['v2v'].append(v2v(x, g).mean(dim=1).double().cpu().numpy())
        res['p2p'].append(p2p(g, x)[0].cpu().numpy())
        mg, mf = bm_gt.forward_vertices(g, f_gt), bm_fit.forward_vertices(x, f_fit)
        bm_gt.check_overflow(); bm_fit.check_overflow()
        d = (mg - mf).abs().double().cpu().numpy()
        for i, k in enumerate(BodyMeasurements.NAMES):
            res[k].append(d[:, i])
    return {k: np.concatenate(v) for k, v in res.items() if v}


def main(input_npz_file, hbw_folder, model_type='smplx', point_reg_gt=DEFAULT_POINT_REG_SMPLX,
         point_reg_fit=DEFAULT_POINT_REG_SMPLX,
         body_measurement_folder=DEFAULT_BODY_MEASUREMENT_FOLDER,
         body_model_folder=DEFAULT_BODY_MODEL_FOLDER):
    result = np.load(input_npz_file)
    res = evaluate(result['image_name'], result['v_shaped'], hbw_folder, model_type, point_reg_gt,
                   point_reg_fit, body_measurement_folder, body_model_folder)
    if model_type == 'smplx':                                        # evaluate_hbw.py:170-186
        print(f'V2V Error: {res["v2v"].mean() * 1000:.0f} mm')
    print(f'P2P-20k Error: {res["p2p"].mean() * 1000:.0f} mm')
    for k in MEAS:
        if k == 'mass':
            print(f'{k} Error: {res[k].mean():.0f} kg')
        else:
            print(f'{k} Error: {res[k].mean() * 1000:.0f} mm')
    return res


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--input-npz-file', dest='input_npz_file', type=str, required=True,
                        help='npz containing labels and body shape parameters.')
    parser.add_argument('--hbw-folder', dest='hbw_folder', type=str, default=DEFAULT_HBW_FOLDER,
                        help='folder with ground truth bodies.')
    parser.add_argument('--model-type', choices=['smpl', 'smplx'], type=str, default='smplx',
                        help='The model type used for body shape prediction. ')
    parser.add_argument('--point-reg-gt', type=str, default=DEFAULT_POINT_REG_SMPLX,
                        help='Point regressor for ground truth SMPL-X mesh.')
    parser.add_argument('--point-reg-fit', type=str, default=DEFAULT_POINT_REG_SMPLX,
                        help='Point regressor for predicted mesh. Topology can be SMPL or SMPL-X.')
    parser.add_argument('--body-measurement-folder', type=str,
                        default=DEFAULT_BODY_MEASUREMENT_FOLDER)
    parser.add_argument('--body-model-folder', type=str, default=DEFAULT_BODY_MODEL_FOLDER)
    a = parser.parse_args()
    main(a.input_npz_file, a.hbw_folder, a.model_type, a.point_reg_gt, a.point_reg_fit,
         a.body_measurement_folder, a.body_model_folder)
