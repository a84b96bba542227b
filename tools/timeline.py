"""Per-phase wall time of one backbone forward from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/timeline.py OUT [--forward -1] [--batch 64]

Kernels are matched to the plan's ops by dispatch order (the host issues the ops in plan order, one
kernel per op), so every kernel gets its op name, lane and epoch (= interval between two joins of
the side streams).  Prints per epoch: wall time (first start .. last end), sum of kernel durations,
the lanes' own sums, and the op list; then totals per phase."""
import argparse
import collections
import csv
import glob
import os.path as osp
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--forward', type=int, default=-1)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--algo', default=None)
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='storage type of the traced run')
    ap.add_argument('--batch', type=int, default=64, help='batch of the traced run (selects the split-K bucket)')
    ap.add_argument('--no-dag', action='store_true', help='the traced run used SHAPY_DAG=0 or --single-stream')
    ap.add_argument('--group', type=int, default=None, help='group_branches of the traced run (default: the product default)')
    args = ap.parse_args()
    import torch                                                   # noqa: F401
    import __graft_entry__ as ge
    net, _ = ge.make_network(device='cpu')
    if args.algo:
        net.backbone.conv_algo = args.algo
    if args.group is not None:
        net.backbone.group_branches = bool(args.group)
    net.backbone._dag_eff = net.backbone.dag and not args.no_dag        # the traced run: eager, multi-stream
    net.backbone._ksplit_eff = net.backbone.ksplit_policy(args.batch) if args.dtype == 'f32' else {}
    plan = net.backbone._build_plan(args.size, args.size, bf16=args.dtype == 'bf16')
    # one kernel per op, except launch groups (one persistent kernel for `group` ops): the group's
    # first op stands for the launch
    ops, skip = [], 0
    for o in plan.ops:
        if skip:
            skip -= 1
            continue
        if o.get('group', 0) > 1:
            skip = o['group'] - 1
            o = dict(o, name=o.get('name', '') + f' [+{skip} grouped]')
        ops.append(o)
    f = glob.glob(args.out + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ks = [r for r in rows if 'conv_' in r['Kernel_Name'] or 'mean_pool' in r['Kernel_Name']]
    n = len(ops)
    # a forward starts with the stem kernel; anything after its n backbone kernels (the SMPL-X
    # blend-shape GEMMs run on the same conv kernel) is not part of the plan
    starts = [i for i, r in enumerate(ks) if 'stem_conv' in r['Kernel_Name']]
    nf = len(starts)
    i0 = starts[args.forward % nf]
    fw = ks[i0:i0 + n]
    assert len(fw) == n and 'mean_pool' in fw[-1]['Kernel_Name'], (len(fw), fw[-1]['Kernel_Name'])
    t0 = min(int(r['Start_Timestamp']) for r in fw)
    epoch, ep_rows = 0, collections.OrderedDict()
    for o, r in zip(ops, fw):
        if o.get('barrier_before'):
            epoch += 1
        ep_rows.setdefault(epoch, []).append((o, r))
    phase_tot = collections.OrderedDict()
    print(f'{nf} forwards in the trace; forward {args.forward % nf}: {n} kernels, wall '
          f'{(max(int(r["End_Timestamp"]) for r in fw) - t0) / 1e3:.1f} us, kernel sum '
          f'{sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw) / 1e3:.1f} us')
    for e, lst in ep_rows.items():
        s = min(int(r['Start_Timestamp']) for _, r in lst)
        t = max(int(r['End_Timestamp']) for _, r in lst)
        lane_sum = collections.defaultdict(float)
        for o, r in lst:
            lane_sum[o['lane']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        nm = lst[0][0].get('name', '') or {0: 'stem', 3: 'meanpool'}.get(lst[0][0]['type'], '?')
        kind = ('branches' if '.branches.' in nm else 'fuse' if '.fuse_layers.' in nm else
                nm.split('.')[0])
        stage = nm.split('.')[0] if kind in ('branches', 'fuse') else ''
        key = f'{stage} {kind}'.strip()
        phase_tot.setdefault(key, [0.0, 0.0, 0])
        phase_tot[key][0] += (t - s) / 1e3
        phase_tot[key][1] += sum(lane_sum.values())
        phase_tot[key][2] += len(lst)
        print(f'epoch {e:3d} @{(s - t0) / 1e3:8.1f} us wall {(t - s) / 1e3:7.1f} us  kernels {len(lst):3d} '
              f'sum {sum(lane_sum.values()):7.1f}  lanes ' +
              ' '.join(f'{l}:{v:.0f}' for l, v in sorted(lane_sum.items())) + f'  first op {nm}')
        if args.verbose:
            for o, r in lst:
                print(f'      lane {o["lane"]} @{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} '
                      f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:6.1f} us  '
                      f'{o.get("name", "")}  {o["Hi"]}x{o["Wi"]} {o["Cin"]}->{o["Cout"]} k{o["ksize"]} '
                      f's{o["stride"]} u{o["ups"]}')
    print('phase totals (wall us, kernel-sum us, launches):')
    for k, v in phase_tot.items():
        print(f'  {k:24s} {v[0]:8.1f} {v[1]:8.1f} {v[2]:4d}')


if __name__ == '__main__':
    main()
