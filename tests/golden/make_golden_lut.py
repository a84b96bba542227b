"""Golden vectors for the dynamic-landmark LUT of SMPL-X (lbs.py:64-110: the yaw of the neck
chain is rounded, clamped to 39 degrees and -- for negative angles -- remapped to rows 40..78 of
the look-up table) from the REAL reference ``SMPLX.forward``.

    python tests/golden/make_golden_lut.py        # build container only (/root/reference)

Eight bodies whose head yaw (a rotation of the root about +y; every other joint near the rest
pose) hits: the rest row, both signs inside the table, the rounding edge next to the clamp
(+39.4 -> 39, -39.6 -> row 78 - ... ), and angles far beyond the clamp on both sides.
Writes ops_golden_lut.npz: yaw_deg, betas, global_rot, body_pose (rotation matrices), joints,
vertices_sub.
"""
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                    # noqa: E402
from oracle import measure as omeasure               # noqa: E402
from shapy_amd.config import merge_config            # noqa: E402
from shapy_amd.utils import synthetic as syn         # noqa: E402

SUB = 7
YAW = np.array([0.0, 25.0, -25.0, 39.4, -39.6, 70.0, -70.0, 100.0, 38.4, -38.6], np.float32)


def main():
    torch.set_num_threads(4)
    ns = ref_loader.load_reference(intersect_fn=omeasure.mesh_to_mesh_forward)
    data_dir = osp.join(ROOT, 'shapy_amd', 'data')
    model_folder = '/tmp/shapy_synth_models'
    syn.write_synthetic_smplx(model_folder, 0)
    cfg = merge_config([osp.join(ROOT, 'configs/b2a_expose_hrnet_demo.yaml')], [
        f'body_model.model_folder={model_folder}',
        'network.smplx.backbone.hrnet.pretrained_path=',
        f'network.smplx.meas_definition_path={data_dir}/measurement_defitions.yaml',
        f'network.smplx.meas_vertices_path={data_dir}/smplx_measurements.yaml',
    ])
    smplx = ns.body_models.SMPLX(osp.join(model_folder, 'smplx'), **cfg.body_model.smplx)
    r = syn.rng_for(0, 'lut')
    B = len(YAW)
    betas = (0.5 * r.standard_normal((B, 10))).astype(np.float32)
    # small random rotations everywhere (axis-angle, 2 degrees), then the yaw on the root
    aa = (r.standard_normal((B, 22, 3)) * np.deg2rad(2.0)).astype(np.float32)
    aa[:, [0, 3, 6, 9, 12, 15]] = 0      # the chain of HEAD_IDX (body_models.py:587): exact yaw
    rot = ns.rotation_utils.batch_rodrigues(torch.from_numpy(aa.reshape(-1, 3))).view(B, 22, 3, 3)
    th = np.deg2rad(YAW.astype(np.float64))
    ry = np.zeros((B, 3, 3), np.float64)
    ry[:, 0, 0] = np.cos(th); ry[:, 0, 2] = np.sin(th)
    ry[:, 1, 1] = 1
    ry[:, 2, 0] = -np.sin(th); ry[:, 2, 2] = np.cos(th)
    rot = rot.numpy().copy()
    rot[:, 0] = ry.astype(np.float32)
    rot_t = torch.from_numpy(rot)
    with torch.no_grad():
        so = smplx(global_rot=rot_t[:, :1], body_pose=rot_t[:, 1:], betas=torch.from_numpy(betas),
                   get_skin=True, return_shaped=True)
    # the LUT row each body used, recomputed with the reference's own function (lbs.py:64-110)
    with torch.no_grad():
        full = torch.cat([rot_t, torch.eye(3).expand(B, 33, 3, 3)], dim=1)   # jaw, eyes, hands
        idx, _ = ns.lbs.find_dynamic_lmk_idx_and_bcoords(
            so['vertices'], full, smplx.dynamic_lmk_faces_idx,
            smplx.dynamic_lmk_bary_coords, smplx.neck_kin_chain)
    lut = smplx.dynamic_lmk_faces_idx.numpy()
    rows = np.array([int(np.where((lut == idx[b].numpy()).all(axis=1))[0][0]) for b in range(B)])
    print('yaw', YAW, '\nLUT rows', rows)
    np.savez(osp.join(HERE, 'ops_golden_lut.npz'), yaw_deg=YAW, betas=betas, rot=rot,
             lut_rows=rows, joints=so['joints']._t.numpy(),
             vertices_sub=so['vertices'].numpy()[:, ::SUB])


if __name__ == '__main__':
    main()
