"""Per-layer error of the Winograd layers against the direct kernel on a probe batch
(HighResolutionNet.calibrate), for the benign synthetic weights and for "wild" ones: BatchNorm
gamma / sigma spread over 10^3 per channel and large positive post-ReLU means.

    python tools/wino_guard_report.py [--wild] [--batch 4] [--budget 2e-5]
"""
import argparse
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_wild(backbone, seed=0, spread=30.0, shift=0.5, damp=0.5):
    """In place: per-channel BatchNorm gain gamma / sqrt(var) log-uniform in [1 / spread, spread]
    (ratio spread^2 ~ 10^3), renormalised per layer to mean square damp^2 so that activations stay
    finite through the depth (features of scale ~30 with the defaults, ~350 with shift 1 / damp 0.6),
    and beta = +shift * |gain| * U(0,1) (positive post-ReLU means: the DC that F(4x4) amplifies)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in backbone.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.num_features
                gain = torch.exp((torch.rand(c, generator=g) * 2 - 1) * float(np.log(spread)))
                gain = damp * gain / gain.pow(2).mean().sqrt()
                var = torch.exp((torch.rand(c, generator=g) * 2 - 1) * float(np.log(10.0)))
                m.running_var.copy_(var)
                m.weight.copy_(gain * torch.sqrt(var + m.eps))
                m.bias.copy_(shift * gain.abs() * torch.rand(c, generator=g))
    backbone.invalidate()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--wild', action='store_true')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--budget', type=float, default=None)
    args = ap.parse_args()
    import __graft_entry__ as ge
    from shapy_amd.utils import synthetic as syn
    net, _ = ge.make_network()
    bb = net.backbone
    if args.wild:
        make_wild(bb)
    x = torch.from_numpy(syn.synthetic_images(args.batch, 224, 3)).cuda()
    bb.wino_guard = False
    rep = bb.calibrate(x, budget=args.budget, demote=False, log=print)
    arr = np.array([(l[2], l[3]) for l in rep['layers']])
    by = {}
    for name, algo, e, em in rep['layers']:
        by.setdefault(algo, []).append((e, em, name))
    for algo, lst in by.items():
        es = np.array([v[0] for v in lst]); ms = np.array([v[1] for v in lst])
        worst = max(lst)
        print(f'{algo:10s} {len(lst):3d} layers: rms-relative error median {np.median(es):.2e} p90 {np.percentile(es, 90):.2e} '
              f'max {es.max():.2e} ({worst[2]}); max-relative median {np.median(ms):.2e} max {ms.max():.2e}')
    rep2 = bb.calibrate(x, budget=args.budget, demote=True, log=print)
    print('demoted:', len(rep2['demoted']), 'of', len(rep2['layers']))
    with torch.no_grad():
        f_guard = bb(x)['concat'].clone()
        bb.layer_algo = {}; bb._calibrated_ver = bb._weights_version()
        f_w4 = bb(x)['concat'].clone()
        bb.conv_algo = 'direct'
        f_dir = bb(x)['concat'].clone()
    sc = float(f_dir.abs().max())
    print(f'features scale {sc:.3g}: |guarded - direct| max {float((f_guard - f_dir).abs().max()):.2e}, '
          f'|unguarded winograd4 - direct| max {float((f_w4 - f_dir).abs().max()):.2e}')


if __name__ == '__main__':
    main()
