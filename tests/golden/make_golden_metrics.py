"""Golden vectors for the evaluator metrics from the reference's own classes
(regressor/human_shape/utils/metrics.py) run on the CPU in this container.

    python tests/golden/make_golden_metrics.py
"""
import importlib
import os
import os.path as osp
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import metrics_inputs                                # noqa: E402
import ref_loader                                    # noqa: E402
from shapy_amd.utils import synthetic as syn         # noqa: E402


def main():
    ref_loader.install_stubs()
    stub = types.ModuleType('human_shape.utils.np_utils')
    stub.np2o3d_pcl = lambda x: x                    # open3d is not installed; unused here
    sys.modules['human_shape.utils.np_utils'] = stub
    M = importlib.import_module('human_shape.utils.metrics')

    I = metrics_inputs.make_inputs()
    est, gt = I['est'], I['gt']
    out = {'est': est, 'gt': gt}
    for kind in ('none', 'translation', 'scale', 'procrustes'):
        out[f'err_{kind}'] = M.PointError(M.build_alignment(kind))(est, gt)
        out[f'aligned_{kind}'] = M.build_alignment(kind)(est, gt)[0]
    joints, joints_gt = I['joints'], I['joints_gt']
    out['err_root'] = M.PointError(M.build_alignment('root', root=[2, 3]))(joints, joints_gt)

    reg_in, reg_tg, tgt = I['reg_in'], I['reg_tg'], I['tgt']
    tmp = tempfile.mkdtemp()
    paths = []
    for name, m in (('in', reg_in), ('tg', reg_tg)):
        p = osp.join(tmp, name + '.pkl')
        with open(p, 'wb') as f:
            pickle.dump(m.tocsr(), f)
        paths.append(p)
    # align=False raises UnboundLocalError in the reference (metrics.py:451-454: ``t`` is only
    # bound when self.align), so only the aligned variant has a reference answer.
    metric = M.v2vhdError(paths[0], paths[1], align=True)
    mean, err = metric(torch.from_numpy(est).double(), torch.from_numpy(tgt).double())
    out['p2p_mean'] = mean.numpy()
    out['p2p_err'] = err.numpy()
    # keep the committed file small: errors are stored for a strided subset of the vertices
    keep = np.arange(0, est.shape[1], 7)
    out['keep'] = keep
    for kind in ('none', 'translation', 'scale', 'procrustes'):
        out[f'err_{kind}'] = out[f'err_{kind}'][:, keep]
        out[f'aligned_{kind}'] = out[f'aligned_{kind}'][:, keep]
    del out['est'], out['gt']                     # regenerated from the seed by the tests
    np.savez_compressed(osp.join(HERE, 'metrics_golden.npz'), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
    print('size', os.path.getsize(osp.join(HERE, 'metrics_golden.npz')))


if __name__ == '__main__':
    main()
