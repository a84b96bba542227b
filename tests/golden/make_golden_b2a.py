"""Golden vector for the B2A polynomial head from the reference's own Polynomial module
(attributes/attributes/attributes_betas/polynomial.py) with seeded weights.

    python tests/golden/make_golden_b2a.py
"""
import importlib.util
import os.path as osp
import sys
import types

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_loader                                    # noqa: E402
from shapy_amd.utils import synthetic as syn         # noqa: E402


def main():
    ref_loader.install_stubs()
    for name in ('attributes.utils', 'attributes.utils.typing'):
        m = types.ModuleType(name)
        m.Tensor = torch.Tensor
        m.Array = np.ndarray
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location(
        'ref_polynomial', osp.join(ref_loader.REF_ROOT, 'attributes', 'attributes',
                                   'attributes_betas', 'polynomial.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    poly = mod.Polynomial(10, 15, degree=2)
    r = syn.rng_for(0, 'b2a')
    W = (r.standard_normal((15, 65)) * 0.3).astype(np.float32)
    b = r.standard_normal(15).astype(np.float32)
    x = r.standard_normal((7, 10)).astype(np.float32)
    with torch.no_grad():
        poly.linear.weight.copy_(torch.from_numpy(W))
        poly.linear.bias.copy_(torch.from_numpy(b))
        y = poly(torch.from_numpy(x)).numpy()
    keys = sorted(poly.state_dict().keys())
    np.savez(osp.join(HERE, 'b2a_golden.npz'), W=W, b=b, x=x, y=y, keys=np.array(keys),
             idx1=poly.indices_001.numpy())
    print(keys, y.shape)


if __name__ == '__main__':
    main()
