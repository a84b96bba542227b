"""Golden vectors at 256 x 256 -- the reference's DEFAULT crop size
(regressor/human_shape/config/datasets_defaults.py:30; neither experiment YAML overrides it), i.e.
the size demo.py feeds the network -- from the REAL reference code, like make_golden.py.

Run in the build container only (needs the read-only reference tree at /root/reference):

    python tests/golden/make_golden_256.py

  hrnet_golden_256.npz      reference HighResolutionNet (CPU), seeded weights, 2 images @256:
                            features + checksums of the intermediate stages
  regressor_golden_256.npz  reference SMPLXRegressor.forward (CPU), B=2 @256, seeded everything

64 / 32 / 16 / 8-pixel maps take other F(4x4) tile counts, launch schedules and workspace
packings than the 56 / 28 / 14 / 7 of the 224 x 224 fixtures.  Weights / images / SMPL-X buffers
are regenerated bit-identically from their seeds (shapy_amd/utils/synthetic.py).
"""
import os.path as osp
import sys
import time

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                    # noqa: E402
from make_golden import SUB, checksums               # noqa: E402
from oracle import measure as omeasure               # noqa: E402
from shapy_amd.config import merge_config            # noqa: E402
from shapy_amd.utils import synthetic as syn         # noqa: E402

SIZE, B = 256, 2


def main():
    torch.set_num_threads(8)
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
    net = ns.body_heads.BODY_HEAD_REGISTRY['SMPLXRegressor'](
        cfg.body_model, network_cfg=cfg.network.smplx, loss_cfg=cfg.losses.body)
    net.eval()
    syn.fill_module_synthetic(net, 0)

    stage_stats = {}

    def hook(name):
        def f(m, i, o):
            if torch.is_tensor(o):
                stage_stats[name] = checksums(o)
            elif isinstance(o, (list, tuple)):
                for q, t in enumerate(o):
                    stage_stats[f'{name}.{q}'] = checksums(t)
        return f
    for n, m in net.backbone.named_modules():
        if n in ('bn1', 'bn2', 'layer1', 'stage2', 'stage3', 'stage4', 'subsample_4',
                 'subsample_3', 'subsample_2', 'conv_layers'):
            m.register_forward_hook(hook(n))
    x = torch.from_numpy(syn.synthetic_images(B, SIZE, 0))
    hg = {}
    tag = f'b{B}_{SIZE}'
    with torch.no_grad():
        t0 = time.time()
        feat = net.backbone(x)['concat']
        print(tag, 'hrnet fwd', time.time() - t0, 's; feat std', feat.std().item())
    hg[tag] = feat.numpy()
    for k, v in stage_stats.items():
        hg[f'{tag}.cs.{k}'] = v
    np.savez(osp.join(HERE, 'hrnet_golden_256.npz'), **hg)

    rg = {}
    with torch.no_grad():
        t0 = time.time()
        out = net(x, None)
        print('full forward', time.time() - t0, 's')
    rg['features'] = out['features'].numpy()
    for i in range(3):
        st = out[f'stage_{i:02d}']
        rg[f'stage{i}_betas'] = st['betas'].numpy()
        rg[f'stage{i}_raw_body_pose'] = st['raw_body_pose'].numpy()
        rg[f'stage{i}_raw_global_rot'] = st['raw_global_rot'].numpy()
        rg[f'stage{i}_camera'] = st['camera'].numpy()
    st = out['stage_02']
    rg['global_rot'] = st['global_rot'].numpy()
    rg['body_pose'] = st['body_pose'].numpy()
    rg['joints'] = st['joints']._t.numpy()
    rg['vertices_sub'] = st['vertices'].numpy()[:, ::SUB]
    rg['v_shaped_sub'] = st['v_shaped'].numpy()[:, ::SUB]
    rg['vertices_cs'] = checksums(st['vertices'])
    rg['v_shaped_cs'] = checksums(st['v_shaped'])
    rg['proj_joints'] = out['proj_joints']._t.numpy() if hasattr(out['proj_joints'], '_t') \
        else out['proj_joints'].numpy()
    rg['cam_scale'] = out['camera_parameters'].scale.numpy()
    for k, v in out['measurements'].items():
        rg['meas_' + k] = v.numpy().astype(np.float32)
        print('regressor measurement', k, rg['meas_' + k])
    np.savez(osp.join(HERE, 'regressor_golden_256.npz'), **rg)
    for f in ('hrnet_golden_256.npz', 'regressor_golden_256.npz'):
        print(f, osp.getsize(osp.join(HERE, f)))


if __name__ == '__main__':
    main()
