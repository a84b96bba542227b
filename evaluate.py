"""Batch inference / evaluation entry point on MI355X -- CLI of regressor/evaluate.py:

    python evaluate.py --exp-cfg configs/b2a_expose_hrnet_eval_shape.yaml --exp-opts ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py \
        --num-gpus 8 --exp-cfg ... --exp-opts ...

The reference initialises ``torch.distributed`` but evaluates on rank 0 only
(regressor/evaluate.py:68-79, human_shape/evaluation.py:641-642).  Here every rank processes a
contiguous shard of the image list (data parallel, no exchange during the forward) and the
per-person results (betas, the five measurements, v_shaped) are all-gathered with RCCL; rank 0
writes ``<output_folder>/<results_folder>/predictions.npz`` in the HBW submission layout
(``image_name``, ``v_shaped``; regressor/hbw_evaluation/test_submission_format.py:4-45) plus
``betas`` and ``measurements``.  The ground-truth metrics of the reference's Evaluator (v2v_t,
p2p-20k, measurement errors; evaluation.py:192-357) run on the GPU through
``shapy_amd.evaluation.Evaluator`` for datasets whose targets carry the ground-truth fields; the
licensed HBW / SSP-3D readers themselves are out of scope.
"""
import logging
import os
import os.path as osp
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)

from shapy_amd import parallel                                  # noqa: E402
from shapy_amd.config import parse_args                         # noqa: E402
from shapy_amd.datasets import OpenPose, batches, crop_and_normalize   # noqa: E402
from shapy_amd.models import build_model                        # noqa: E402
from shapy_amd.utils.checkpointer import Checkpointer           # noqa: E402

logger = logging.getLogger('shapy_amd')
MEAS = ('mass', 'height', 'chest', 'waist', 'hips')


@torch.no_grad()
def main(exp_cfg):
    if not torch.cuda.is_available():
        logger.error('No GPU is available!')
        sys.exit(3)                                    # evaluate.py:48-51
    local_rank = int(os.environ.get('LOCAL_RANK', exp_cfg.local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda')
    distributed = exp_cfg.num_gpus > 1
    rank, world = 0, 1
    if distributed:
        rank, world = parallel.init_distributed(exp_cfg.backend)

    model = build_model(exp_cfg)['network'].to(device=device)
    output_folder = osp.expandvars(exp_cfg.output_folder)
    Checkpointer(model, save_dir=osp.join(output_folder, exp_cfg.checkpoint_folder),
                 pretrained=exp_cfg.pretrained, rank=rank).load_checkpoint()
    model = model.eval()

    part_key = exp_cfg.get('part_key', 'pose')
    part_cfg = exp_cfg.datasets[part_key]
    transf = part_cfg.get('transforms', {})
    crop_size = transf.get('crop_size', 256)
    dataset = OpenPose(split='val', **part_cfg.get('openpose', {}))

    names, betas, meas, v_shaped = [], [], [], []
    for batch in batches(dataset, exp_cfg.datasets.batch_size, rank, world):
        imgs, targets = [b[0] for b in batch], [b[1] for b in batch]
        x = crop_and_normalize(imgs, [t.get_field('center') for t in targets],
                               [t.get_field('scale') for t in targets], crop_size,
                               transf.get('mean', (0.485, 0.456, 0.406)),
                               transf.get('std', (0.229, 0.224, 0.225)), device=device)
        out = model(x, targets)
        st = out['stage_02']
        names += [t.get_field('fname') for t in targets]
        betas.append(st['betas'])
        v_shaped.append(st['v_shaped'])
        if 'measurements' in out:
            meas.append(torch.stack([out['measurements'][k] for k in MEAS], dim=1))
    cat = lambda xs, shape: torch.cat(xs) if xs else torch.zeros(shape, device=device)
    betas = parallel.gather_variable(cat(betas, (0, 10)))
    meas = parallel.gather_variable(cat(meas, (0, 5)))
    v_shaped = parallel.gather_variable(cat(v_shaped, (0, model.model.num_verts, 3)))
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, names)
        names = [n for part in gathered for n in part]
        dist.barrier()
    if rank == 0:
        res_dir = osp.join(output_folder, exp_cfg.results_folder)
        os.makedirs(res_dir, exist_ok=True)
        np.savez_compressed(osp.join(res_dir, 'predictions.npz'), image_name=np.array(names),
                            v_shaped=v_shaped.cpu().numpy(), betas=betas.cpu().numpy(),
                            measurements=meas.cpu().numpy(), measurement_names=np.array(MEAS))
        logger.info('wrote %d predictions to %s', len(names), res_dir)
    if distributed:
        dist.destroy_process_group()
    return len(names)


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(message)s')
    cfg = parse_args()
    cfg.is_training = False
    main(cfg)
