"""Lower bounds per phase of the backbone forward next to the measured wall times.

For every epoch group of the plan (stem + layer1, stage 2, stage 3, stage 4, head): the FLOPs the
matrix cores execute / the nominal dense f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) and the
bytes every conv must move (input + output + residual + weights, each once) / 4.7 TB/s (mixed
read / write rate reached by the HBM-bound layers) -- the larger of the two is the phase's floor.

    python tools/phase_floors.py [--batch 64] [--timeline profiles/r03x_timeline_multistream_dag.txt]
"""
import argparse
import os.path as osp
import re
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--timeline', default='profiles/r03x_timeline_multistream_dag.txt')
    args = ap.parse_args()
    import __graft_entry__ as ge
    from shapy_amd import _lib
    net, _ = ge.make_network(device='cpu')
    bb = net.backbone
    bb._dag_eff = True
    P = bb._build_plan(224, 224)
    B = args.batch
    phases = {}

    def phase_of(name):
        if name.startswith(('conv2', 'layer1')) or name == '':
            return 'stem + conv2 + layer1'
        if name.startswith(('transition1', 'stage2')):
            return 'transition1 + stage 2'
        if name.startswith(('transition2', 'stage3')):
            return 'transition2 + stage 3'
        if name.startswith(('transition3', 'stage4')):
            return 'transition3 + stage 4'
        return 'head (subsample, 5 Bottlenecks, mean)'
    for o in P.ops:
        ph = phases.setdefault(phase_of(o.get('name', '')), dict(flop=0.0, bytes=0.0, n=0))
        ph['n'] += 1
        if o['type'] == _lib.OP_MEANPOOL:
            ph['bytes'] += 4.0 * B * o['Hi'] * o['Wi'] * o['Cin']
            continue
        cc = o['Cout'] * o['Cin']
        if o['type'] == _lib.OP_STEM:
            macs = o['Ho'] * o['Wo'] * cc * 9
            ph['bytes'] += 4.0 * B * (o['Hi'] * o['Wi'] * 3 + o['Ho'] * o['Wo'] * o['Cout'])
            continue                                         # (VALU kernel: no matrix-core work)
        if o.get('wino_off', -1) >= 0 and o['tile'] & _lib.TILE_WINO4:
            macs = 36 * -(-o['Ho'] // 4) * -(-o['Wo'] // 4) * cc
            wbytes = 36 * cc * 4
        elif o.get('wino_off', -1) >= 0:
            macs = 16 * -(-o['Ho'] // 2) * -(-o['Wo'] // 2) * cc
            wbytes = 16 * cc * 4
        else:
            macs = o['Ho'] * o['Wo'] * cc * o['ksize'] ** 2
            wbytes = cc * o['ksize'] ** 2 * 4
        ph['flop'] += 2.0 * B * macs
        out_px = o['Ho'] * o['Wo'] * o['ups'] ** 2
        ph['bytes'] += 4.0 * B * (o['Hi'] * o['Wi'] * o['Cin'] + out_px * o['Cout'] *
                                  (2 if o['resb'] is not None else 1)) + wbytes
    walls = {}
    if osp.exists(args.timeline):
        agg = {'stem + conv2 + layer1': 0.0, 'transition1 + stage 2': 0.0, 'transition2 + stage 3': 0.0,
               'transition3 + stage 4': 0.0, 'head (subsample, 5 Bottlenecks, mean)': 0.0}
        for ln in open(args.timeline):
            m = re.match(r'epoch\s+\d+ @\s*[\d.]+ us wall\s+([\d.]+) us .* first op (\S*)', ln)
            if m:
                nm = m.group(2) if m.group(2) != '?' else ''
                agg[phase_of(nm if not nm.startswith(('subsample', 'conv_layers')) else 'head')] += float(m.group(1))
        walls = agg
    print(f'{"phase":40s} {"launches":>8s} {"MFMA floor":>11s} {"HBM floor":>10s} {"measured":>9s}   (ms, B = {B})')
    tot = [0.0, 0.0, 0.0]
    for k, v in phases.items():
        t_m, t_h = v['flop'] / 157.3e12 * 1e3, v['bytes'] / 4.7e12 * 1e3
        w = walls.get(k, float('nan')) / 1e3
        tot[0] += max(t_m, t_h); tot[1] += w if w == w else 0.0
        print(f'{k:40s} {v["n"]:8d} {t_m:11.2f} {t_h:10.2f} {w:9.2f}')
    print(f'{"sum of the phase floors":40s} {"":8s} {tot[0]:11.2f} {"":10s} {tot[1]:9.2f}')


if __name__ == '__main__':
    main()
