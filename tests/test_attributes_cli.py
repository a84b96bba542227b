"""B2A attribute head and the virtual-measurements CLI."""
import os.path as osp
import sys

import numpy as np
import pytest
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(HERE)


def test_polynomial_layout_matches_reference(golden_dir):
    from shapy_amd.models.attributes import Polynomial
    g = np.load(osp.join(golden_dir, 'b2a_golden.npz'))
    poly = Polynomial(10, 15, degree=2)
    assert sorted(poly.state_dict().keys()) == list(g['keys'])
    assert np.array_equal(poly.indices_001.numpy(), g['idx1'])
    assert poly.coeff_size == 65


@pytest.mark.gpu
def test_b2a_kernel_vs_reference_golden(golden_dir, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from shapy_amd.models.attributes import B2A, Polynomial
    g = np.load(osp.join(golden_dir, 'b2a_golden.npz'))
    poly = Polynomial(10, 15).cuda()
    with torch.no_grad():
        poly.linear.weight.copy_(torch.from_numpy(g['W']))
        poly.linear.bias.copy_(torch.from_numpy(g['b']))
    y = poly(torch.from_numpy(g['x']).cuda()).cpu().numpy()
    assert np.abs(y - g['y']).max() < 1e-5
    # Lightning-style checkpoint round trip (state_dict keys 'b2a.*')
    ck = tmp_path / 'last.ckpt'
    torch.save({'state_dict': {'b2a.' + k: v for k, v in poly.state_dict().items()},
                'hyper_parameters': {}}, ck)
    m = B2A.load_from_checkpoint(str(ck)).cuda()
    assert np.abs(m(torch.from_numpy(g['x']).cuda()).cpu().numpy() - g['y']).max() < 1e-5


@pytest.mark.gpu
def test_regressor_attributes_routing(tmp_path):
    """use_b2a with both checkpoints present: attributes [B,15], zero rows for unknown gender
    (iterative_regressor.py:761-776)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    sys.path.insert(0, ROOT)
    from shapy_amd.config import merge_config
    from shapy_amd.datasets import Target
    from shapy_amd.models import build_model
    from shapy_amd.models.attributes import Polynomial
    from shapy_amd.utils import synthetic as syn
    torch.manual_seed(0)
    for name in ('m', 'f'):
        p = Polynomial(10, 15)
        torch.save({'state_dict': {'b2a.' + k: v for k, v in p.state_dict().items()}},
                   tmp_path / f'{name}.ckpt')
    syn.write_synthetic_smplx('/tmp/shapy_synth_models', 0)
    cfg = merge_config([osp.join(ROOT, 'configs/b2a_expose_hrnet_demo.yaml')], [
        'body_model.model_folder=/tmp/shapy_synth_models',
        'network.smplx.backbone.hrnet.pretrained_path=', 'network.smplx.compute_measurements=False',
        f'network.smplx.b2a_males_checkpoint={tmp_path}/m.ckpt',
        f'network.smplx.b2a_females_checkpoint={tmp_path}/f.ckpt'])
    net = build_model(cfg)['network']
    assert net.use_b2a
    syn.fill_module_synthetic(net, 0)
    net = net.cuda().eval()
    x = torch.from_numpy(syn.synthetic_images(3, 64, 0)).cuda()
    targets = [Target(gender='male'), Target(), Target(gender='F')]
    with torch.no_grad():
        out = net(x, targets)
    a = out['attributes']
    assert a.shape == (3, 15)
    assert (a[1] == 0).all() and a[0].abs().sum() > 0 and a[2].abs().sum() > 0
    betas = out['stage_02']['betas']
    assert torch.allclose(a[0], net.b2a_males(betas[:1])[0])
    assert torch.allclose(a[2], net.b2a_females(betas[2:3])[0])


@pytest.mark.gpu
def test_virtual_measurements_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    sys.path.insert(0, osp.join(ROOT, 'measurements'))
    import virtual_measurements as vm
    from oracle import body_np, measure
    from shapy_amd.utils import synthetic as syn
    syn.write_synthetic_smplx('/tmp/shapy_synth_models', 0)
    r = syn.rng_for(1, 'vm')
    (tmp_path / 'in').mkdir()
    betas = r.standard_normal(10).astype(np.float32)
    np.savez(tmp_path / 'in' / 'a.npz', betas=betas)
    data = osp.join(ROOT, 'shapy_amd', 'data')
    res = vm.main(str(tmp_path / 'in'), str(tmp_path / 'out'),
                  osp.join(data, 'measurement_defitions.yaml'),
                  osp.join(data, 'smplx_measurements.yaml'), '/tmp/shapy_synth_models')
    model = syn.make_synthetic_smplx(0)
    v = (model['v_template'][None] + body_np.blend_shapes(betas[None], model['shapedirs'][:, :, :10]))
    lm = measure.load_landmarks(osp.join(data, 'measurement_defitions.yaml'),
                                osp.join(data, 'smplx_measurements.yaml'))
    ref = measure.body_measurements(v[:, model['f']], lm)
    for k in ('mass', 'height', 'chest', 'waist', 'hips'):
        assert abs(res['a.npz'][k] - float(ref[k][0])) < (2e-4 if k == 'mass' else 1e-5), k
    assert osp.exists(tmp_path / 'out' / 'a_measurements.json')
