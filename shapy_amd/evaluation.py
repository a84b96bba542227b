"""Evaluator: shape metrics on the GPU, data-parallel over ranks.

Mirror of the metric half of regressor/human_shape/evaluation.py: ``build_metric_utilities``
(:570-637), ``compute_metric`` and its helpers ``_compute_v2v`` / ``_compute_p2p`` /
``_compute_measurement_error`` (:192-357) and the accumulation loop of ``run`` (:640-767).
Differences, all deliberate:
  * the metrics are HIP launches on the tensors the forward pass left in HBM
    (``shapy_amd.utils.metrics``); the reference copies every output to the host and uses numpy;
  * ``run`` is sharded: the reference returns immediately on rank > 0 (:641-642); here every rank
    evaluates the batches it is handed and the per-metric (sum, count) pairs are all-reduced,
    so the logged means equal the single-process ones;
  * tensorboard summaries, rendering, BMI histograms (:420-560, :668-760) and the keypoint
    metrics (mpjpe / mpjpe14, which need the licensed 3-D joint annotations) are out of scope.
"""
import logging
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from .utils.metrics import PointError, build_alignment, v2vhdError

logger = logging.getLogger('shapy_amd')


def _target_vertices(target, key):
    v = target.get_field(key)
    v = getattr(v, 'vertices', v)                    # data/structures Vertices or a raw tensor
    return torch.as_tensor(v)


class Evaluator(object):
    def __init__(self, exp_cfg, rank=0, distributed=False, part_key='body'):
        self.rank = rank
        self.distributed = distributed
        self.part_key = part_key
        self.metrics = self.build_metric_utilities(exp_cfg, part_key)

    def build_metric_utilities(self, exp_cfg, part_key):
        """evaluation.py:570-637 without the keypoint metrics."""
        eval_cfg = exp_cfg.get('evaluation', {}).get(part_key, {})
        metrics = {
            'v2v': {name: PointError(build_alignment(name)) for name in eval_cfg.get('v2v', ())},
            'v2v_t': {name: PointError(build_alignment(name))
                      for name in eval_cfg.get('v2v_t', ())},
            'measurements': None,
        }
        p2p_t_cfg = dict(eval_cfg.get('p2p_t', {}))
        if p2p_t_cfg.get('input_point_regressor_path') and \
                p2p_t_cfg.get('target_point_regressor_path'):
            metrics['p2p_t'] = v2vhdError(**p2p_t_cfg)
        return metrics

    # ---- per-batch metrics -------------------------------------------------------------
    def _compute_v2v(self, model_output, targets, metric_align_dicts, vertex_key='vertices',
                     metric_name='v2v', **extra_args):
        """evaluation.py:192-224."""
        idx = [ii for ii, t in enumerate(targets) if t.has_field(vertex_key)]
        if len(idx) < 1:
            return {}
        est = model_output[vertex_key]
        gt = torch.stack([_target_vertices(targets[ii], vertex_key) for ii in idx]).to(est.device)
        # (the reference does not index the estimate with gt_verts_indices either: it assumes
        # every target of the batch carries the field, evaluation.py:215-219)
        return {f'{name}_{metric_name}': align(est, gt)
                for name, align in metric_align_dicts.items()}

    def _compute_p2p(self, model_output, targets, metric, vertex_key='v_shaped',
                     metric_name='p2p_t', **extra_args):
        """evaluation.py:227-262."""
        idx = [ii for ii, t in enumerate(targets) if t.has_field(vertex_key)]
        if len(idx) < 1:
            return {}
        est = model_output[vertex_key]
        gt = torch.stack([_target_vertices(targets[ii], vertex_key) for ii in idx]).to(est.device)
        diff, _ = metric(est, gt)
        return {metric_name: diff}

    def _compute_measurement_error(self, model_output, targets):
        """evaluation.py:265-296: |gt - est| for the samples with a positive ground truth."""
        est_measurements = model_output.get('measurements', {})
        out = {}
        for name, val in est_measurements.items():
            if not torch.is_tensor(val):
                continue
            idx, gt = [], []
            for ii, t in enumerate(targets):
                if t.has_field(name) and float(t.get_field(name)) > 0:
                    idx.append(ii)
                    gt.append(float(t.get_field(name)))
            if len(idx) < 1:
                continue
            gt = torch.tensor(gt, dtype=torch.float64, device=val.device)
            sel = val.reshape(val.shape[0], -1)[:, 0][torch.tensor(idx, device=val.device)]
            out[name] = (gt - sel.double()).abs()
        return out

    def compute_metric(self, model_output, targets, metrics, **extra_args):
        """evaluation.py:303-357."""
        out = {}
        for metric_name, metric in metrics.items():
            if metric_name == 'v2v':
                out.update(self._compute_v2v(model_output, targets, metric, vertex_key='vertices',
                                             **extra_args))
            elif metric_name == 'v2v_t':
                out.update(self._compute_v2v(model_output, targets, metric, metric_name='v2v_t',
                                             vertex_key='v_shaped', **extra_args))
            elif metric_name == 'measurements':
                out.update(self._compute_measurement_error(model_output, targets))
            elif metric_name == 'p2p_t':
                out.update(self._compute_p2p(model_output, targets, metric, metric_name='p2p_t',
                                             vertex_key='v_shaped', **extra_args))
            else:
                raise ValueError(f'Unsupported metric: {metric_name}')
        return out

    # ---- accumulation over a (sharded) dataset ----------------------------------------------
    def reduce(self, metric_values):
        """name -> mean over every element of every rank (the reference's
        ``np.mean(np.concatenate(values))``, evaluation.py:753-757), times 1000 (mm)."""
        local = {k: (float(sum(v.double().sum() for v in vals)), sum(v.numel() for v in vals))
                 for k, vals in metric_values.items()}
        if self.distributed and dist.is_initialized():
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, local)
        else:
            gathered = [local]
        total = defaultdict(lambda: [0.0, 0])
        for part in gathered:
            for k, (s, n) in part.items():
                total[k][0] += s
                total[k][1] += n
        return {k: 1000.0 * s / n for k, (s, n) in sorted(total.items()) if n > 0}

    @torch.no_grad()
    def run(self, model, batches, device, metric_names=('v2v_t', 'p2p_t', 'measurements'), step=0,
            dset_name='dataset'):
        """``batches`` yields ``(images, targets)`` of THIS rank's shard.  Returns the reduced
        metric means in mm (identical on every rank)."""
        model.eval()
        metric_values = defaultdict(list)
        # one batch of lookahead: the next batch is uploaded before this one runs, and a network that can use it
        # (SMPLXRegressor: accepts_next_images) runs its stem + layer1 under this batch's head (bit-identical
        # outputs; models/backbone/prefetch.py)
        it = iter(batches)
        ahead = getattr(model, 'accepts_next_images', False)
        nxt = next(it, None)
        nxt = None if nxt is None else (nxt[0].to(device=device), nxt[1])
        while nxt is not None:
            (images, targets), nxt = nxt, next(it, None)
            nxt = None if nxt is None else (nxt[0].to(device=device), nxt[1])
            extra = {'next_images': nxt[0]} if ahead and nxt is not None else {}
            model_output = model(images, targets, device=device, **extra)
            num_stages = model_output.get('num_stages', 1)
            stage_n_out = model_output.get(f'stage_{num_stages - 1:02d}', {})
            cur = self.compute_metric(
                stage_n_out, targets,
                metrics={m: self.metrics[m] for m in metric_names if m in self.metrics})
            for key, value in cur.items():
                metric_values[key].append(value)
        means = self.reduce(metric_values)
        if self.rank == 0:
            for name, val in means.items():
                logger.info('[%06d] %s, %s: %.4f (mm)', step, dset_name, name, val)
        return means


def to_numpy(metric_dict):
    """name -> numpy array, the type the reference's compute_metric returns."""
    return {k: v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            for k, v in metric_dict.items()}
