"""Evaluator metrics: oracle vs the reference's golden, HIP vs oracle/golden.

Tolerances: the aligned point errors are float32 quantities of O(0.01-1 m); the reference
reduces in float32 (numpy), the kernel in float64 -> 2e-6 absolute.  P2P is float64 on both
sides -> 1e-11.
"""
import os
import os.path as osp
import sys

import numpy as np
import pytest
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, osp.join(ROOT, 'tests', 'golden'))
import metrics_inputs                                             # noqa: E402
from oracle import metrics_np                                     # noqa: E402
from shapy_amd.datasets.structures import Target                  # noqa: E402

KINDS = ('none', 'translation', 'scale', 'procrustes')


@pytest.fixture(scope='module')
def inputs():
    return metrics_inputs.make_inputs()


@pytest.fixture(scope='module')
def golden(golden_dir):
    return np.load(osp.join(golden_dir, 'metrics_golden.npz'))


# ---------------------------------------------------------------- CPU: oracle pinned to the reference
@pytest.mark.parametrize('kind', KINDS)
def test_oracle_alignment_vs_reference(inputs, golden, kind):
    keep = golden['keep']
    ali = metrics_np.align(inputs['est'], inputs['gt'], kind)
    err = metrics_np.point_error(ali, inputs['gt'])
    np.testing.assert_allclose(ali[:, keep], golden[f'aligned_{kind}'], atol=2e-6, rtol=0)
    np.testing.assert_allclose(err[:, keep], golden[f'err_{kind}'], atol=2e-6, rtol=0)


def test_oracle_p2p_vs_reference(inputs, golden):
    mean, err = metrics_np.p2p_error(inputs['reg_in'], inputs['reg_tg'], inputs['est'],
                                     inputs['tgt'])
    np.testing.assert_allclose(err, golden['p2p_err'], atol=1e-12, rtol=0)
    np.testing.assert_allclose(mean, golden['p2p_mean'], atol=1e-12, rtol=0)


def test_oracle_measurement_error():
    est = {'mass': np.array([60., 70., 80.]), 'height': np.array([1.6, 1.7, 1.8])}
    gt = {'mass': np.array([61., 0., 78.]), 'height': np.array([1.65, 1.7, -1.])}
    out = metrics_np.measurement_error(est, gt)
    np.testing.assert_allclose(out['mass'], [1., 2.])
    np.testing.assert_allclose(out['height'], [0.05, 0.0], atol=1e-12)


def _reduce_worker(rank, world, port, q):
    import torch.distributed as dist
    from shapy_amd.config.node import ConfigNode
    from shapy_amd.evaluation import Evaluator
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ev = Evaluator(ConfigNode({}), rank=rank, distributed=True)
    g = torch.Generator().manual_seed(0)
    full = torch.rand(7, 11, generator=g, dtype=torch.float64)
    lo, hi = (0, 4) if rank == 0 else (4, 7)
    vals = {'translation_v2v_t': [full[lo:hi]]}
    if rank == 1:                                        # a key only one rank has
        vals['mass'] = [torch.tensor([2.0, 4.0], dtype=torch.float64)]
    q.put((rank, ev.reduce(vals), float(full.mean()) * 1000))
    dist.destroy_process_group()


def test_evaluator_run_looks_one_batch_ahead():
    """Evaluator.run uploads the NEXT batch before running the current one and hands it to networks that declare
    `accepts_next_images` (SMPLXRegressor: its stem + layer1 then run under the current batch's head); every batch is
    evaluated once, in order, the last one without a successor; other networks are called as the reference calls them."""
    from shapy_amd.config.node import ConfigNode
    from shapy_amd.evaluation import Evaluator
    ev = Evaluator(ConfigNode({}))
    ev.compute_metric = lambda stage_out, targets, metrics: {'v2v_t': stage_out['err']}
    data = [(torch.full((2, 3, 4, 4), float(i)), {'id': i}) for i in range(4)]
    calls = []

    class Net(torch.nn.Module):
        accepts_next_images = True

        def forward(self, images, targets, device=None, next_images=None):
            calls.append((float(images[0, 0, 0, 0]), targets['id'],
                          None if next_images is None else float(next_images[0, 0, 0, 0])))
            return {'num_stages': 1, 'stage_00': {'err': images[:, 0, 0, 0].double() / 1000.0}}

    means = ev.run(Net(), iter(data), 'cpu', metric_names=())
    assert calls == [(0.0, 0, 1.0), (1.0, 1, 2.0), (2.0, 2, 3.0), (3.0, 3, None)]
    assert abs(means['v2v_t'] - 1.5) < 1e-12                  # mean of 0, 1, 2, 3 (x 1000: mm)

    class Plain(torch.nn.Module):                              # no such attribute: the reference's call
        def forward(self, images, targets, device=None):
            calls.append(targets['id'])
            return {'num_stages': 1, 'stage_00': {'err': images[:, 0, 0, 0].double()}}

    del calls[:]
    ev.run(Plain(), data[:1], 'cpu', metric_names=())
    assert calls == [0] and ev.run(Plain(), [], 'cpu', metric_names=()) == {}


def test_evaluator_reduce_gloo_world2():
    """The sharded accumulation equals the single-process mean (evaluation.py:753-757)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    for _, means, want in res:
        assert abs(means['translation_v2v_t'] - want) < 1e-9
        assert abs(means['mass'] - 3000.0) < 1e-9


# ---------------------------------------------------------------- GPU: HIP vs oracle and golden
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize('kind', KINDS)
def test_hip_aligned_error_vs_golden_and_oracle(inputs, golden, kind):
    from shapy_amd.utils import metrics as M
    est = torch.from_numpy(inputs['est']).cuda()
    gt = torch.from_numpy(inputs['gt']).cuda()
    align = M.build_alignment(kind)
    err = M.PointError(align)(est, gt).cpu().numpy()
    ali = align(est, gt)[0].cpu().numpy()
    keep = golden['keep']
    np.testing.assert_allclose(err[:, keep], golden[f'err_{kind}'], atol=2e-6, rtol=0)
    np.testing.assert_allclose(ali[:, keep], golden[f'aligned_{kind}'], atol=2e-6, rtol=0)
    np.testing.assert_allclose(err, metrics_np.aligned_point_error(inputs['est'], inputs['gt'], kind),
                               atol=2e-6, rtol=0)
    mean = align.error(est, gt, per_point=False).cpu().numpy()
    np.testing.assert_allclose(mean, err.mean(1), atol=1e-6, rtol=0)
    assert M.point_error(est[0], gt[0]).shape == (1, est.shape[1])   # 2-D input -> batch of one


@gpu
def test_hip_root_alignment_vs_golden(inputs, golden):
    from shapy_amd.utils import metrics as M
    pe = M.PointError(M.build_alignment('root', root=[2, 3]))
    err = pe(inputs['joints'], inputs['joints_gt']).cpu().numpy()     # numpy in, like the reference
    np.testing.assert_allclose(err, golden['err_root'], atol=1e-6, rtol=0)


@gpu
def test_hip_procrustes_reflection_and_exact_recovery():
    """det(R) = +1 must hold when the best orthogonal map is a reflection (metrics.py:146-149),
    and a pure similarity transform is recovered exactly."""
    from shapy_amd.utils import metrics as M
    r = np.random.default_rng(3)
    est = r.standard_normal((4, 500, 3)).astype(np.float32)
    gt = est.copy()
    gt[..., 0] *= -1                                     # mirrored target
    got = M.PointError(M.ProcrustesAlignment())(est, gt).cpu().numpy()
    # errors here are O(5) and the three singular values nearly coincide, so the float32
    # arithmetic of the reference itself moves the answer by ~1e-5 (float32 vs float64 oracle);
    # the kernel accumulates in float64 and must sit on the float64 oracle.
    want64 = metrics_np.aligned_point_error(est.astype(np.float64), gt.astype(np.float64), 'procrustes')
    np.testing.assert_allclose(got, want64, atol=3e-6, rtol=0)
    np.testing.assert_allclose(got, metrics_np.aligned_point_error(est, gt, 'procrustes'), atol=3e-5)
    q, _ = np.linalg.qr(r.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    gt = (1.7 * est @ q.T + np.array([0.3, -2.0, 5.0])).astype(np.float32)
    got = M.PointError(M.ProcrustesAlignment())(est, gt).cpu().numpy()
    assert got.max() < 5e-6


@gpu
def test_hip_p2p_vs_golden_and_oracle(inputs, golden):
    from shapy_amd.utils import metrics as M
    metric = M.v2vhdError(input_point_regressor=inputs['reg_in'],
                          target_point_regressor=inputs['reg_tg'], align=True)
    est = torch.from_numpy(inputs['est']).cuda()          # float32 estimate, float64 target
    tgt = torch.from_numpy(inputs['tgt']).cuda()
    mean, err = metric(est, tgt)
    assert err.dtype == torch.float64
    np.testing.assert_allclose(err.cpu().numpy(), golden['p2p_err'], atol=1e-11, rtol=0)
    np.testing.assert_allclose(mean.cpu().numpy(), golden['p2p_mean'], atol=1e-11, rtol=0)
    # align=False crashes in the reference (unbound ``t``); the oracle states the evident intent
    metric.align = False
    mean, err = metric(est, tgt)
    wm, we = metrics_np.p2p_error(inputs['reg_in'], inputs['reg_tg'], inputs['est'], inputs['tgt'],
                                  do_align=False)
    np.testing.assert_allclose(err.cpu().numpy(), we, atol=1e-11, rtol=0)
    np.testing.assert_allclose(mean.cpu().numpy(), wm, atol=1e-11, rtol=0)


@gpu
def test_hip_p2p_pickle_paths(inputs, tmp_path):
    """Constructor of the reference: pickled scipy.sparse files (metrics.py:378-392)."""
    import pickle
    from shapy_amd.utils import metrics as M
    paths = []
    for name in ('reg_in', 'reg_tg'):
        p = tmp_path / f'{name}.pkl'
        with open(p, 'wb') as f:
            pickle.dump(inputs[name].tocoo(), f)          # COO on disk, like the released files
        paths.append(str(p))
    metric = M.v2vhdError(paths[0], paths[1])
    mean, _ = metric(inputs['est'][:2], inputs['tgt'][:2])
    want, _ = metrics_np.p2p_error(inputs['reg_in'], inputs['reg_tg'], inputs['est'][:2],
                                   inputs['tgt'][:2])
    np.testing.assert_allclose(mean.cpu().numpy(), want, atol=1e-11, rtol=0)


@gpu
def test_hip_metrics_full_batch_properties():
    """bs 256 x 10,475 vertices: translation invariance and idempotence of the alignment."""
    from shapy_amd.utils import metrics as M
    g = torch.Generator(device='cuda').manual_seed(0)
    est = torch.randn(256, 10475, 3, device='cuda', generator=g)
    shift = torch.randn(256, 1, 3, device='cuda', generator=g)
    tr = M.TranslationAlignment()
    assert float(M.PointError(tr)(est, est + shift).max()) < 2e-6
    noise = 0.01 * torch.randn(256, 10475, 3, device='cuda', generator=g)
    a = M.PointError(tr)(est, est + noise)
    b = M.PointError(tr)(est + shift, est + noise)        # same error wherever the estimate sits
    assert float((a - b).abs().max()) < 2e-6
    for kind in ('scale', 'procrustes'):
        al = M.build_alignment(kind)
        once = al(est, 1.3 * est + shift)[0]
        twice = al(once, 1.3 * est + shift)[0]
        assert float((once - twice).abs().max()) < 1e-5
        assert float(M.point_error(once, 1.3 * est + shift).max()) < 1e-5


@gpu
def test_evaluator_compute_metric_vs_oracle(inputs):
    from shapy_amd.config.node import ConfigNode
    from shapy_amd.evaluation import Evaluator, to_numpy
    from shapy_amd.utils import metrics as M
    cfg = ConfigNode({'evaluation': {'body': {'v2v_t': ['scale', 'translation'],
                                              'v2v': ['procrustes']}}})
    ev = Evaluator(cfg)
    ev.metrics['p2p_t'] = M.v2vhdError(input_point_regressor=inputs['reg_in'],
                                       target_point_regressor=inputs['reg_in'])
    B = inputs['est'].shape[0]
    est = torch.from_numpy(inputs['est']).cuda()
    mass = torch.tensor([60., 70., 80., 90., 100.]).cuda()
    out = {'v_shaped': est, 'vertices': est, 'measurements': {'mass': mass}}
    gt_mass = [61., 0., 78., 95., 100.]
    targets = [Target(v_shaped=torch.from_numpy(inputs['gt'][b]),
                      vertices=torch.from_numpy(inputs['gt'][b]), mass=gt_mass[b]) for b in range(B)]
    got = to_numpy(ev.compute_metric(out, targets, ev.metrics))
    for kind, key in (('scale', 'scale_v2v_t'), ('translation', 'translation_v2v_t'),
                      ('procrustes', 'procrustes_v2v')):
        want = metrics_np.aligned_point_error(inputs['est'], inputs['gt'], kind)
        np.testing.assert_allclose(got[key], want, atol=2e-6, rtol=0)
    want, _ = metrics_np.p2p_error(inputs['reg_in'], inputs['reg_in'], inputs['est'], inputs['gt'])
    np.testing.assert_allclose(got['p2p_t'], want, atol=1e-11, rtol=0)
    np.testing.assert_allclose(got['mass'], [1., 2., 5., 0.], atol=1e-5)
    with pytest.raises(ValueError):
        ev.compute_metric(out, targets, {'mpjpe': {}})
    means = ev.reduce({k: [torch.from_numpy(v)] for k, v in got.items()})
    assert abs(means['mass'] - 2000.0) < 1e-2


@gpu
def test_evaluate_hbw_cli_vs_oracle(tmp_path, capsys):
    """hbw_evaluation/evaluate_hbw.py (reference CLI, evaluate_hbw.py:61-187) on a synthetic
    HBW tree: V2V / P2P-20k / measurement errors against the numpy + C oracle."""
    import pickle
    import scipy.sparse as sp
    ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
    sys.path.insert(0, osp.join(ROOT, 'hbw_evaluation'))
    import evaluate_hbw as eh
    from oracle import measure as om
    from shapy_amd.utils import synthetic as syn
    faces, meshes = syn.load_topology()
    V = meshes.shape[1]
    r = np.random.default_rng(3)
    # ground truth: 3 subjects in two splits; 5 predictions (two images of one subject)
    (tmp_path / 'hbw' / 'smplx' / 'val').mkdir(parents=True)
    (tmp_path / 'hbw' / 'smplx' / 'test').mkdir(parents=True)
    subj = {('val', '012'): meshes[0], ('val', '033'): 0.5 * (meshes[1] + meshes[2]),
            ('test', '007'): 1.03 * meshes[3]}
    for (split, sid), v in subj.items():
        np.save(tmp_path / 'hbw' / 'smplx' / split / f'{sid}.npy', v.astype(np.float64))
    labels = ['val/012_55/Photos/a.png', 'val/012_55/Photos/b.png', 'val/033_10/Photos/c.png',
              'test/007_01/Photos/d.png', 'val/033_10/Photos/e.png']
    gts = np.stack([subj[(l.split('/')[0], l.split('/')[1].split('_')[0])] for l in labels]).astype(np.float32)
    fits = (gts * r.uniform(0.97, 1.03, (5, 1, 1)) + r.normal(0, 2e-3, gts.shape) +
            r.normal(0, 0.05, (5, 1, 3))).astype(np.float32)
    np.savez(tmp_path / 'pred.npz', image_name=np.array(labels), v_shaped=fits)
    P = 500
    rows = np.repeat(np.arange(P), 3)
    reg = sp.csr_matrix((r.dirichlet(np.ones(3), P).reshape(-1), (rows, r.integers(0, V, 3 * P))),
                        shape=(P, V))
    with open(tmp_path / 'reg.pkl', 'wb') as f:
        pickle.dump(reg, f)
    (tmp_path / 'models' / 'smplx').mkdir(parents=True)
    np.savez(tmp_path / 'models' / 'smplx' / 'SMPLX_NEUTRAL.npz', f=faces)
    data = osp.join(ROOT, 'shapy_amd', 'data')
    res = eh.main(str(tmp_path / 'pred.npz'), str(tmp_path / 'hbw'), 'smplx', str(tmp_path / 'reg.pkl'),
                  str(tmp_path / 'reg.pkl'), data, str(tmp_path / 'models'))
    printed = capsys.readouterr().out
    assert 'V2V Error:' in printed and 'P2P-20k Error:' in printed and 'mass Error:' in printed
    want_v2v = metrics_np.aligned_point_error(fits, gts, 'translation').mean(axis=1)
    np.testing.assert_allclose(res['v2v'], want_v2v, atol=2e-6, rtol=0)
    want_p2p, _ = metrics_np.p2p_error(reg, reg, gts, fits)
    np.testing.assert_allclose(res['p2p'], want_p2p, atol=1e-9, rtol=0)
    lm = om.load_landmarks(osp.join(data, 'measurement_defitions.yaml'),
                           osp.join(data, 'smplx_measurements.yaml'))
    mg, mf = om.body_measurements(gts[:, faces], lm), om.body_measurements(fits[:, faces], lm)
    for k in ('height', 'chest', 'waist', 'hips', 'mass'):
        np.testing.assert_allclose(res[k], np.abs(mg[k].astype(np.float64) - mf[k]),
                                   atol=2e-4 if k == 'mass' else 3e-6, rtol=0, err_msg=k)
