"""Per HighResolutionModule of a verbose timeline (tools/timeline.py --verbose): when each branch lane
finishes, how late each lane's first branch conv starts after the lane's previous kernel, and how long the
fuse phase runs behind the last branch -- the numbers behind DESIGN section 8.

    python tools/module_tails.py profiles/r04o_timeline_multistream_dag_verbose.txt
"""
import collections
import re
import sys


def main():
    rows = []
    for line in open(sys.argv[1]):
        m = re.match(r'\s+lane (\d) @\s*([\d.]+)\s+([\d.]+) us\s+(\S+)', line)
        if m:
            rows.append((int(m[1]), float(m[2]), float(m[2]) + float(m[3]), m[4]))
    mods = collections.OrderedDict()
    for lane, t0, t1, name in rows:
        m = re.match(r'(stage\d\.\d)\.(branches|fuse_layers)\.(\d)', name)
        if m:
            mods.setdefault(m[1], []).append((lane, t0, t1, m[2], int(m[3]), name))
    prev_end = collections.defaultdict(float)          # lane -> end of its last kernel before the module
    order = sorted(rows, key=lambda r: r[1])
    print('module     lane: first branch conv starts (+ idle behind the lane\'s previous kernel), branch ends | '
          'fuse phase: last branch end -> module end')
    for mod, ops in mods.items():
        br = [o for o in ops if o[3] == 'branches']
        fu = [o for o in ops if o[3] != 'branches']
        t_first = min(o[1] for o in br)
        cells = []
        for lane in sorted({o[0] for o in br}):
            mine = [o for o in br if o[0] == lane]
            start = min(o[1] for o in mine)
            before = [r[2] for r in order if r[0] == lane and r[2] <= start + 1e-6]
            idle = start - max(before) if before else 0.0
            cells.append(f'{lane}: {start - t_first:6.0f} (+{idle:4.0f}) .. {max(o[2] for o in mine) - t_first:6.0f}')
        last_branch = max(o[2] for o in br)
        end = max([o[2] for o in fu] + [last_branch])
        print(f'{mod}  ' + ' | '.join(cells) + f' | fuse tail {end - last_branch:5.0f} us, module {end - t_first:6.0f} us')


if __name__ == '__main__':
    main()
