"""Input pre-processing: host logic + oracle vs the reference's own
functions (golden), and the HIP crop/resize/normalise kernel vs the oracle."""
import os.path as osp

import numpy as np
import pytest
import torch

from oracle import preprocess_np
from shapy_amd.datasets import OpenPose
from shapy_amd.datasets.keypoints import crop_window

HERE = osp.dirname(osp.abspath(__file__))
SAMPLES = osp.join(HERE, 'golden', 'samples')


@pytest.fixture(scope='module')
def dataset():
    return OpenPose(data_folder=SAMPLES, img_folder='images', keyp_folder='openpose',
                    body_thresh=0.05, hand_thresh=0.2, head_thresh=0.3, use_face_contour=True)


def test_boxes_windows_and_oracle_crop_match_reference(dataset, golden_dir):
    g = np.load(osp.join(golden_dir, 'preprocess_golden.npz'))
    assert len(dataset) == 3
    for i in range(len(dataset)):
        img, tgt = dataset[i]
        name = tgt.get_field('fname').split('.')[0]
        assert np.array_equal(tgt.get_field('center'), g[f'{name}.center'])
        assert float(tgt.get_field('scale')) == float(g[f'{name}.scale'])
        assert float(tgt.get_field('bbox_size')) == float(g[f'{name}.bbox_size'])
        imgf = np.clip(img.astype(np.float32) / 255.0, 0, 1)
        for res in (224, 256):
            win = crop_window(tgt.get_field('center'), tgt.get_field('scale'), [res, res])
            assert np.array_equal(win, g[f'{name}.window{res}'])
            crop = preprocess_np.crop(imgf, win, res)
            assert crop.shape == (res, res, 3)
            assert np.array_equal(crop[::8, ::8], g[f'{name}.crop{res}_sub'])
            np.testing.assert_allclose(crop.astype(np.float64).sum(), g[f'{name}.crop{res}_cs'][0],
                                       rtol=1e-9)


def test_resize_restatement_basic_properties():
    r = np.random.default_rng(0)
    src = r.random((37, 53, 3)).astype(np.float32)
    same = preprocess_np.resize_bilinear_cv2(src, (53, 37))
    assert np.array_equal(same, src)                       # identity when sizes match
    up = preprocess_np.resize_bilinear_cv2(np.full((5, 7, 3), 0.25, np.float32), (64, 64))
    assert np.allclose(up, 0.25)
    # exact 2x downscale of a linear ramp samples pixel-pair midpoints
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :, None], (2, 1, 1))
    down = preprocess_np.resize_bilinear_cv2(ramp, (4, 1))
    assert np.allclose(down[0, :, 0], [0.5, 2.5, 4.5, 6.5])


@pytest.mark.gpu
@pytest.mark.parametrize('res', [224, 256])
def test_hip_crop_kernel_vs_oracle(dataset, res):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from shapy_amd.datasets import crop_and_normalize
    items = [dataset[i] for i in range(len(dataset))]
    imgs = [it[0] for it in items]
    centers = [it[1].get_field('center') for it in items]
    scales = [it[1].get_field('scale') for it in items]
    out = crop_and_normalize(imgs, centers, scales, res)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == (3, 3, res, res)
    for i in range(3):
        ref = preprocess_np.preprocess(imgs[i], crop_window(centers[i], scales[i], [res, res]), res)
        err = np.abs(out[i] - ref).max()
        assert err < 2e-6, err


@pytest.mark.gpu
def test_demo_entry_point_end_to_end(tmp_path):
    """demo.py on the shipped sample images with a synthetic checkpoint: npz keys of the
    reference, values equal to a direct forward, checkpoint round trip."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sys
    root = osp.dirname(HERE)
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    import demo
    from shapy_amd.utils.checkpointer import Checkpointer
    net, _ = ge.make_network(seed=3)
    exp = tmp_path / 'exp'
    Checkpointer(net, save_dir=str(exp / 'checkpoints')).save_checkpoint('best_checkpoint')
    data = osp.join(root, 'shapy_amd', 'data')
    cfg, a = demo.parse([
        '--exp-cfg', osp.join(root, 'configs', 'b2a_expose_hrnet_demo.yaml'),
        '--output-folder', str(tmp_path / 'out'), '--save-params', 'true', '--save-mesh', 'true',
        '--exp-opts', f'output_folder={exp}', 'body_model.model_folder=/tmp/shapy_synth_models',
        'network.smplx.backbone.hrnet.pretrained_path=',
        f'network.smplx.meas_definition_path={data}/measurement_defitions.yaml',
        f'network.smplx.meas_vertices_path={data}/smplx_measurements.yaml',
        f'datasets.pose.openpose.data_folder={SAMPLES}', 'datasets.pose.openpose.img_folder=images',
        'datasets.pose.openpose.keyp_folder=openpose', 'datasets.batch_size=2'])
    n = demo.main(cfg, demo_output_folder=a.output_folder, save_params=a.save_params,
                  save_mesh=a.save_mesh, split=a.split)
    assert n == 2                                  # 3 people in batches of 2
    d = np.load(osp.join(a.output_folder, 'img_00.npz'), allow_pickle=True)
    ref_keys = {'fname', 'joints', 'vertices', 'v_shaped', 'faces', 'global_rot', 'raw_global_rot',
                'body_pose', 'raw_body_pose', 'betas', 'camera', 'measurements', 'proj_joints',
                'shift_x', 'shift_y', 'transl', 'focal_length_in_mm', 'focal_length_in_px',
                'center', 'sensor_width'}
    assert set(d.files) == ref_keys
    assert d['vertices'].shape == (10475, 3) and d['joints'].shape == (123, 3)
    assert set(d['measurements'].item()) == {'mass', 'height', 'chest', 'waist', 'hips'}
    assert osp.getsize(osp.join(a.output_folder, 'img_00.ply')) > 10475 * 12
    # the checkpoint was really loaded: a direct forward with the seed-3 network agrees
    from shapy_amd.datasets import OpenPose, crop_and_normalize
    ds = OpenPose(data_folder=SAMPLES, img_folder='images', keyp_folder='openpose',
                  **{k: v for k, v in cfg.datasets.pose.openpose.items()
                     if k in ('body_thresh', 'hand_thresh', 'head_thresh', 'use_face_contour')})
    img, tgt = ds[0]
    x = crop_and_normalize([img], [tgt.get_field('center')], [tgt.get_field('scale')], 256)
    with torch.no_grad():
        out = net(x, [tgt])
    assert np.abs(out['stage_02']['betas'][0].cpu().numpy() - d['betas']).max() < 1e-5
