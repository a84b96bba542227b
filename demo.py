"""SHAPY regressor demo on MI355X -- drop-in for regressor/demo.py (same CLI):

    python demo.py --exp-cfg configs/b2a_expose_hrnet_demo.yaml --datasets openpose \
        --output-folder out --save-params true --save-mesh true \
        --exp-opts output_folder=../data/trained_models/shapy/SHAPY_A part_key=pose \
        datasets.pose.openpose.data_folder=../samples datasets.pose.openpose.img_folder=images \
        datasets.pose.openpose.keyp_folder=openpose datasets.batch_size=1 \
        datasets.pose_shape_ratio=1.0

For every person found in the OpenPose keypoint files it writes ``<name>.npz`` with the keys of
the reference (demo.py:337-353: the stage_02 dict + the blender camera of
weak_persp_to_blender) and optionally ``<name>.ply``.  Renderings (--save-vis; pyrender) are out
of scope.  With several processes (torchrun) the image list is sharded contiguously over the
ranks; nothing is exchanged.
"""
import argparse
import logging
import os
import os.path as osp
import sys
import time
from collections import defaultdict

import numpy as np
import torch

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)

from shapy_amd.config import merge_config                      # noqa: E402
from shapy_amd.datasets import OpenPose, batches, crop_and_normalize   # noqa: E402
from shapy_amd.models import build_model                       # noqa: E402
from shapy_amd.models.body_models import KeypointTensor        # noqa: E402
from shapy_amd.utils.checkpointer import Checkpointer          # noqa: E402
from shapy_amd.utils.mesh_io import write_ply                  # noqa: E402

logger = logging.getLogger('shapy_amd')


def weak_persp_to_blender(targets, camera_scale, camera_transl, H, W, sensor_width=36,
                          focal_length=5000):
    """demo.py:71-106."""
    camera_scale = camera_scale.detach().cpu().numpy()
    camera_transl = camera_transl.detach().cpu().numpy()
    output = defaultdict(list)
    for ii, target in enumerate(targets):
        orig_bbox_size = target.get_field('orig_bbox_size')
        bbox_center = target.get_field('orig_center')
        z = 2 * focal_length / (camera_scale[ii] * orig_bbox_size)
        output['shift_x'].append(-(bbox_center[0] / W[ii] - 0.5))
        output['shift_y'].append((bbox_center[1] - 0.5 * H[ii]) / W[ii])
        output['transl'].append([camera_transl[ii, 0].item(), camera_transl[ii, 1].item(), z.item()])
        output['focal_length_in_mm'].append(focal_length / W[ii] * sensor_width)
        output['focal_length_in_px'].append(focal_length)
        output['center'].append(bbox_center)
        output['sensor_width'].append(sensor_width)
    return {k: np.array(v) for k, v in output.items()}


@torch.no_grad()
def main(exp_cfg, demo_output_folder='demo_output', focal_length=5000, sensor_width=36,
         save_vis=False, save_params=False, save_mesh=False, split='test'):
    if not torch.cuda.is_available():
        logger.error('No GPU is available!')
        sys.exit(3)                                   # demo.py:136-139
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    device = torch.device('cuda')
    if save_vis:
        logger.warning('--save-vis needs the pyrender renderers of the reference: skipped')
    output_folder = osp.expandvars(exp_cfg.output_folder)
    os.makedirs(demo_output_folder, exist_ok=True)

    model = build_model(exp_cfg)['network'].to(device=device)
    checkpoint_folder = osp.join(output_folder, exp_cfg.checkpoint_folder)
    Checkpointer(model, save_dir=checkpoint_folder, pretrained=exp_cfg.pretrained,
                 rank=rank).load_checkpoint()
    model = model.eval()

    part_key = exp_cfg.get('part_key', 'pose')
    part_cfg = exp_cfg.datasets[part_key]
    transf = part_cfg.get('transforms', {})
    crop_size = transf.get('crop_size', 256)
    mean, std = transf.get('mean', (0.485, 0.456, 0.406)), transf.get('std', (0.229, 0.224, 0.225))
    names = part_cfg.splits[split]
    if list(names) != ['openpose']:
        raise NotImplementedError(f'only the openpose dataset is supported, got {list(names)}')
    dataset = OpenPose(split=split, **part_cfg.get('openpose', {}))
    logger.info('%d people in %s', len(dataset), dataset.img_folder)

    def prepared():
        for batch in batches(dataset, exp_cfg.datasets.batch_size, rank, world):
            imgs = [b[0] for b in batch]
            targets = [b[1] for b in batch]
            body_imgs = crop_and_normalize(imgs, [t.get_field('center') for t in targets],
                                           [t.get_field('scale') for t in targets], crop_size, mean,
                                           std, device=device)
            yield imgs, targets, body_imgs

    # one batch ahead (not in the reference loop, demo.py:307-353): the next batch's crops exist before this
    # batch's forward is issued, and the network runs their stem + layer1 under this batch's head
    # (SMPLXRegressor.forward(next_images=), models/backbone/prefetch.py; same outputs)
    look_ahead = getattr(model, 'accepts_next_images', False)
    total_time, cnt = 0.0, 0
    it = prepared()
    cur = next(it, None)
    while cur is not None:
        imgs, targets, body_imgs = cur
        nxt = next(it, None)
        torch.cuda.synchronize()
        start = time.perf_counter()
        if look_ahead and nxt is not None and nxt[2].shape == body_imgs.shape:
            out = model(body_imgs, targets, next_images=nxt[2])
        else:
            out = model(body_imgs, targets)
        torch.cuda.synchronize()
        if getattr(model, 'compute_measurements', False):
            model.body_measurements.check_overflow()
        total_time += time.perf_counter() - start
        cnt += 1
        cur = nxt

        cam = out['camera_parameters']
        hd_params = weak_persp_to_blender(
            targets, cam.scale, cam.translation, H=[i.shape[0] for i in imgs],
            W=[i.shape[1] for i in imgs], sensor_width=sensor_width, focal_length=focal_length)
        stage = out['stage_02']
        vertices = stage['vertices'].detach().cpu().numpy()
        for idx, target in enumerate(targets):
            fname = target.get_field('fname')
            stem = fname.split('.')[0]
            if save_mesh:
                write_ply(osp.join(demo_output_folder, f'{stem}.ply'),
                          vertices[idx] + hd_params['transl'][idx], stage['faces'])
            if save_params:
                params = dict(fname=fname)
                for key, val in stage.items():
                    if isinstance(val, KeypointTensor):
                        val = val._t[idx].detach().cpu().numpy()
                    elif torch.is_tensor(val):
                        val = val.detach().cpu().numpy()[idx]
                    elif isinstance(val, dict):          # measurements: name -> [B]
                        val = {k: v[idx].item() for k, v in val.items()}
                    params[key] = val
                for key, val in hd_params.items():
                    params[key] = val[idx].item() if np.isscalar(val[idx]) else val[idx]
                np.savez_compressed(osp.join(demo_output_folder, f'{stem}.npz'), **params)
    if cnt:
        logger.info('Average inference time: %f', total_time / cnt)
    return cnt


def parse(argv=None):
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                description='SMPL-X regressor demo (MI355X)')
    flag = lambda x: x.lower() in ['true']
    p.add_argument('--exp-cfg', type=str, dest='exp_cfgs', nargs='+')
    p.add_argument('--output-folder', dest='output_folder', default='demo_output', type=str)
    p.add_argument('--datasets', nargs='+', default=['openpose'], type=str)
    p.add_argument('--show', default=False, type=flag)
    p.add_argument('--pause', default=-1, type=float)
    p.add_argument('--exp-opts', default=[], dest='exp_opts', nargs='*')
    p.add_argument('--focal-length', dest='focal_length', type=float, default=5000)
    p.add_argument('--save-vis', dest='save_vis', default=False, type=flag)
    p.add_argument('--save-mesh', dest='save_mesh', default=False, type=flag)
    p.add_argument('--save-params', dest='save_params', default=False, type=flag)
    p.add_argument('--split', default='test', type=str, choices=['train', 'test', 'val'])
    a = p.parse_args(argv)
    cfg = merge_config(a.exp_cfgs or [], a.exp_opts)
    cfg.is_training = False
    for part_key in ['pose', 'shape']:                    # demo.py:423-430
        splits = cfg.datasets.get(part_key, {}).get('splits', {})
        if splits:
            splits['train'], splits['val'], splits['test'] = [], [], []
    cfg.datasets[cfg.get('part_key', 'pose')].splits[a.split] = list(a.datasets)
    return cfg, a


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(message)s')
    cfg, a = parse()
    main(cfg, demo_output_folder=a.output_folder, focal_length=a.focal_length,
         save_vis=a.save_vis, save_mesh=a.save_mesh, save_params=a.save_params, split=a.split)
