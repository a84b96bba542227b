"""Headline benchmark: images/sec of the SHAPY hot path (HRNet-W48 + iterative regressor +
SMPL-X + virtual measurements) on synthetic 224x224 crops, batch 64 per GPU, float32.

    python bench.py --gpus 1 --steps 20 --warmup 5          # BASELINE configs[1] (default)
    python bench.py --gpus 8                                # spawns its own 8 RCCL ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --workload measurements --meshes 1000   # BASELINE configs[3]
    python bench.py --workload smplx --batch 64             # SMPL-X layer alone (LBS evidence)
    python bench.py --workload bvh --meshes 1000            # LBVH path of the intersection op

One "step" of the default workload = one full forward of the regressor on one batch that is
already resident in HBM (BASELINE.json configs[1]: "HRNet-W48 + SMPL-X head, random-init
weights, 224x224 bs=64 fp32 on 1xMI355X"); with N > 1 every rank runs its own shard (weak
scaling: 64 images per GPU) and the predicted betas are all-gathered with RCCL once per step on
a side stream (joined when the next step issues its gather, shapy_amd/parallel.py).  Consecutive steps are
software-pipelined (--pipeline on, the default): every step hands the next batch to the network, whose stem +
layer1 then run under this batch's head on one of the executor's side streams (bit-identical outputs;
shapy_amd/models/backbone/prefetch.py); the timed region holds exactly `steps` whole forwards of work.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      the roofline of the dominant kernel family.  regressor: MFMA; `achieved` = the
                FLOPs the matrix cores EXECUTE (Winograd layers: 36 products per 4x4 tile / 16 per
                2x2 tile and channel pair) / time per forward -- HIP events around the backbone call on
                the launch stream with --pipeline off, the whole step period with --pipeline on (the
                events then bracket only a part of a forward; `roofline.duration` says which) --,
                peak 157.3 TFLOP/s (f32 MFMA, dense),
                so frac <= 1 by construction; the direct-convolution-equivalent rate (2 x
                18,466,524,160 MAC per image) is `algorithmic_equiv_tflops`.
                measurements / smplx / bvh: HBM, algorithmic bytes
                (376.6 KB per mesh; 65.4 MB constants + 254.4 KB per body) / time
                of the launch group, peak 8 TB/s
  parity        (regressor) error of the LAST timed batch against the CPU oracle, computed after
                the timed region: the "betas L2 vs CPU" half of BASELINE.json's metric
  cpu_baseline  the CPU oracle ("port": torch-CPU restatement of the reference) timed on this
                host's cores on a bounded sample of the same workload
  rccl_ranks / per_rank   (N > 1) number of RCCL ranks and each rank's own images/s
  also          (default N = 1 run) compact sub-records of the OTHER BASELINE configurations, timed
                after the headline's timed region: configs[0] (SMPL-X layer), configs[2]'s per-GPU
                shard (bf16, bs 32), configs[3] (1,000 meshes), the LBVH path, the 256 x 256 crop, the
                headline one forward at a time, the opt-in bf16x6 head GEMMs; the same values as flat
                also_<key>_{value,ms,roofline_frac} scalars at the top level
"""
import argparse
import json
import os
import os.path as osp
import subprocess
import sys
import time

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

CONV_FLOP_PER_IMAGE_224 = 2 * 18_466_524_160       # SURVEY.md 8(d), counted from the reference
F32_MFMA_PEAK_TFLOPS = 157.3                       # MI355X_MICROARCH.md, dense f32 MFMA
BF16_MFMA_PEAK_TFLOPS = 2500.0                     # dense bf16 MFMA (not the 2:1-sparse figure)
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md (6,290 GB/s achievable)
MEASURE_BYTES_PER_MESH = 10475 * 12 + 20908 * 12   # v_shaped f32 + int32 faces = 376.6 KB (8d)
SMPLX_CONST_BYTES = 65.4e6                         # posedirs + weights + J_regressor + ... (8d)
SMPLX_BYTES_PER_BODY = 2 * 125_700 + 3_000         # vertices + v_shaped written, params read


def baseline_metric():
    """The metric string of BASELINE.json, verbatim."""
    try:
        with open(osp.join(ROOT, 'BASELINE.json')) as fh:
            return json.load(fh)['metric']
    except (OSError, KeyError, ValueError):
        return 'images/sec whole-node (HRNet+SMPL-X fwd), 224×224 bs=64; betas L2 vs CPU'


def _f32_plan(net, size, batch=None):
    """The backbone's float32 op list at this size: the compiled engine's own plan when the run
    is float32 (the engine of the CURRENT layer_algo: the Winograd guard may have rebuilt the plan
    after demoting layers; of the batch bucket that takes / does not take the bf16x6 head GEMMs), a freshly
    built one otherwise (the bf16 / f32x6 plans pad channels)."""
    bb = net.backbone
    cur = tuple(sorted(getattr(bb, 'layer_algo', {}).items()))
    x6 = None if batch is None else 0 < getattr(bb, 'x6_gemm_min_batch', 0) <= batch
    best = None
    for key, eng in bb._engine.items():
        if key[0] == size and key[1] == size and key[3] == 'f32' and key[4] == bb.conv_algo \
                and (x6 is None or key[13] == x6):
            if cur in key:
                return eng['plan']
            best = eng['plan']
    return best if best is not None else bb._build_plan(size, size)


def launches_per_forward(plan):
    """Kernel launches of one backbone forward: every op is one launch, except that the members of a
    launch group (ShapyOp.group = n on the first of n ops) share ONE persistent launch."""
    n, skip = 0, 0
    for o in plan.ops:
        if skip:
            skip -= 1
            continue
        n += 1
        skip = max(0, o.get('group', 0) - 1)
    return n


def conv_flop_per_image(net, size, plan=None):
    plan = plan or _f32_plan(net, size)              # f32 plan: algorithmic (unpadded) MACs
    macs = sum(o['Ho'] * o['Wo'] * o['Cout'] * o['Cin'] * o['ksize'] ** 2
               for o in plan.ops if o['type'] in (0, 1))          # CONV, STEM (not MEANPOOL / FUSEADD)
    return 2 * macs


#: a layer on the bf16x6 kernel issues 6 bf16 MFMAs per float32 multiply-add; the matrix-pipe TIME it needs, in
#: FLOPs of the f32 pipe: 6 x (f32 peak / bf16 peak) per FLOP
X6_F32_PIPE_EQUIV = 6.0 * F32_MFMA_PEAK_TFLOPS / BF16_MFMA_PEAK_TFLOPS


def executed_mfma_flop_per_image(net, size, plan=None, batch=None, parts=False):
    """FLOPs the matrix cores actually execute per image with the backbone's conv_algo: direct
    layers as counted above, Winograd F(2x2,3x3) layers 16 products per 2x2 tile and channel pair,
    F(4x4,3x3) layers 36 per 4x4 tile (whole tiles: partly filled edge tiles count in full).  Layers on the
    bf16x6 kernel (SHAPY_TILE_X6: the head's wide 1x1 GEMMs from bs 64) run on the bf16 matrix cores: they count
    with the matrix-pipe time they need, expressed in f32-pipe FLOPs (x 0.3775), so that
    achieved / f32 peak stays "least matrix-pipe time / measured time" <= 1.  parts: (f32-core FLOPs, bf16-core
    FLOPs of the x6 layers) instead."""
    from shapy_amd import _lib
    plan = plan or _f32_plan(net, size, batch)
    macs = macs_x6 = 0
    for o in plan.ops:
        if o['type'] not in (0, 1):
            continue
        cc = o['Cout'] * o['Cin']
        if o.get('wino_off', -1) >= 0 and o['tile'] & _lib.TILE_WINO4:
            macs += 36 * -(-o['Ho'] // 4) * -(-o['Wo'] // 4) * cc
        elif o.get('wino_off', -1) >= 0:
            macs += 16 * -(-o['Ho'] // 2) * -(-o['Wo'] // 2) * cc
        elif o['tile'] & _lib.TILE_X6:
            macs_x6 += o['Ho'] * o['Wo'] * cc * o['ksize'] ** 2
        else:
            macs += o['Ho'] * o['Wo'] * cc * o['ksize'] ** 2
    if parts:
        return 2 * macs, 12 * macs_x6
    return 2 * macs + 2 * macs_x6 * X6_F32_PIPE_EQUIV


def pmc_traffic(batch, size, dtype='f32', algo='direct', any_algo=False):
    """HBM bytes per backbone forward from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_hbm_traffic*.json; FETCH_SIZE and WRITE_SIZE need separate passes and cannot
    be collected from inside this process).  None when the workload differs.  any_algo: the newest
    pass of the same batch / size / dtype whatever its conv_algo (reported as context only)."""
    import glob
    for f in sorted(glob.glob(osp.join(ROOT, 'profiles', '*_pmc_hbm_traffic*.json')), reverse=True):
        with open(f) as fh:
            d = json.load(fh)
        h = d.get('hbm_bytes_per_backbone_forward', {})
        same_algo = h.get('algo', 'direct') == (algo if dtype == 'f32' else h.get('algo', 'direct'))
        if (h.get('batch') == batch and h.get('size') == size and h.get('dtype', 'f32') == dtype
                and (same_algo or any_algo)):
            out = {'bytes_as_reported': h['as_reported'], 'bytes_fetch_x2_corrected':
                   h['fetch_x2_corrected'], 'conv_algo': h.get('algo', 'direct'),
                   'source': osp.relpath(f, ROOT)}
            m = d.get('mfma')
            out['pmc_plan'] = d.get('plan', 'bench.py --single-stream')
            if m and m.get('GRBM_GUI_ACTIVE_sum_over_8_xcd'):
                # matrix-core busy share of all SIMD cycles over the PROFILED pass (dispatches serialised by the
                # counter collection): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over 8 XCDs x 128 SIMDs)
                out['mfma_busy_frac_of_simd_cycles'] = (
                    m['SQ_VALU_MFMA_BUSY_CYCLES_sum'] / (m['GRBM_GUI_ACTIVE_sum_over_8_xcd'] * 128.0))
            if m and m.get('busy_cycles_per_forward'):
                out['mfma_busy_cycles_per_forward'] = m['busy_cycles_per_forward']
            return out
    return None


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            return next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), '')
    except OSError:
        return ''


def cpu_baseline_and_parity(x_np, out, size):
    """CPU oracle on the LAST timed batch (rank 0, after the timed region): its wall time is the
    all-cores CPU baseline at the headline batch size, its result is the parity reference.
    SURVEY.md 8(d): warm-up 1, median of 5 full batches (a bounded sample: ~25 s of host time on the GPU box's
    64 cores; SHAPY_CPU_BASELINE_REPS overrides), time.perf_counter, same inputs / weights, at the
    headline batch AND at bs = 4, all cores and ONE thread (the reference pins its pools to one
    thread, demo.py:432)."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    state = ge.oracle_state(0)
    cores = min(torch.get_num_threads(), 64)      # more threads only add scheduling noise here
    torch.set_num_threads(cores)
    n = x_np.shape[0]
    reps = int(os.environ.get('SHAPY_CPU_BASELINE_REPS', '5'))

    def timed(xs, k):
        ts, last = [], None
        for _ in range(k):
            t0 = time.perf_counter()
            last = ge.oracle_forward(xs, state=state)
            ts.append(time.perf_counter() - t0)
        return ts, last
    ge.oracle_forward(x_np[:min(n, 4)], state=state)              # untimed warm-up pass
    times, ref = timed(x_np, reps)                                # the whole batch, every sample
    dt = float(np.median(times))
    n4 = min(n, 4)
    times4, _ = timed(x_np[:n4], reps)
    dt4 = float(np.median(times4))
    torch.set_num_threads(1)
    n1 = min(n, 3)
    ge.oracle_forward(x_np[:1], state=state)
    times1, _ = timed(x_np[:n1], 3)
    dt1 = float(np.median(times1))
    torch.set_num_threads(cores)
    fmt = lambda ts: ', '.join(f'{t:.2f}' for t in ts)            # noqa: E731
    base = {'value': n / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'cpu_model': cpu_model(), 'host_logical_cpus': os.cpu_count(),
            'sample': f'{n} images ({size}x{size}, one batch of {n}) through the CPU oracle '
                      f'(torch-CPU HRNet + numpy SMPL-X + C intersection + scipy hull): median of '
                      f'{reps} full-batch passes after one warm-up = {dt:.2f} s (samples {fmt(times)} s)',
            'batch_4': {'value': n4 / dt4, 'unit': 'images/sec', 'cores': cores,
                        'sample': f'one batch of {n4} images: median of {reps} = {dt4:.2f} s '
                                  f'(samples {fmt(times4)} s)'},
            'single_thread': {'value': n1 / dt1, 'unit': 'images/sec', 'cores': 1,
                              'sample': f'{n1} images (one batch), median of 3 = {dt1:.2f} s with '
                                        'torch.set_num_threads(1) (the reference pins its pools to '
                                        '1 thread, demo.py:432)'}}
    st, rs = out['stage_02'], ref['stages'][-1]
    betas = st['betas'].float().cpu().numpy()
    db = betas - rs['betas']
    dv = st['vertices'].float().cpu().numpy() - rs['vertices']
    par = {'n_images': int(n), 'reference': 'CPU oracle (float32, same seeded weights and images)',
           'betas_l2': float(np.sqrt((db * db).sum(axis=1)).mean()),
           'betas_l2_max': float(np.sqrt((db * db).sum(axis=1)).max()),
           'betas_maxabs': float(np.abs(db).max()),
           'vertices_maxabs': float(np.abs(dv).max()),
           'features_maxabs': float(np.abs(out['features'].float().cpu().numpy() -
                                           ref['features']).max()),
           'tolerance': 1e-4}
    if 'measurements' in out and 'measurements' in ref:
        par['measurements_maxabs'] = {
            k: float(np.abs(out['measurements'][k].float().cpu().numpy() - ref['measurements'][k]).max())
            for k in ('mass', 'height', 'chest', 'waist', 'hips')}
    return base, par


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through
    torch.distributed.run, the launch convention of regressor/evaluate.py:68-79 (env://)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), osp.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SHAPY_BENCH_SPAWNED='1')
    return subprocess.call(cmd, env=env)


STUB = {'on': False}      # --cpu-stub (tests/test_host_cpu.py): gloo ranks, stub forward, no GPU


def barrier():
    """Barrier of the control plane.  gloo (default, N > 1): a host barrier -- every caller brackets it
    with device_sync(); nccl (--control-backend nccl): an RCCL barrier on this rank's own GPU."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == 'gloo':
        dist.barrier()
    else:
        dist.barrier(device_ids=[torch.cuda.current_device()])


def control_device():
    """Where the control plane's small tensors (timings) live: host for gloo, this GPU for nccl."""
    import torch.distributed as dist
    return 'cpu' if (STUB['on'] or not dist.is_initialized() or dist.get_backend() == 'gloo') else 'cuda'


def device_sync():
    import torch
    if not STUB['on']:
        torch.cuda.synchronize()


class _HostEvent:
    """perf_counter stand-in for a HIP event (--cpu-stub only)."""

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def hip_events(n):
    import torch
    if STUB['on']:
        return [_HostEvent() for _ in range(n)], [_HostEvent() for _ in range(n)]
    return ([torch.cuda.Event(enable_timing=True) for _ in range(n)],
            [torch.cuda.Event(enable_timing=True) for _ in range(n)])


def make_stub_network(size):
    """--cpu-stub: a CPU module with the regressor's interface (backbone hook points, output dict
    with stage_02.betas) whose "betas" are a fixed linear function of the images -- enough to run
    the whole N-rank control flow of this file (self-spawn, rendezvous, barriers, the deferred
    all-gather, max-over-ranks timing, the JSON line) on the gloo backend without a GPU."""
    import torch
    import torch.nn as nn

    class _Backbone(nn.Module):
        conv_algo, use_graph, graph_max_batch, wino4_min_hw = 'stub', False, 0, 0
        multi_stream, compute_dtype, tile_flags, _engine = False, 'f32', 0, {}

        def forward(self, x):
            return {'concat': x.mean(dim=(2, 3))}

    class _Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = _Backbone()
            g = torch.Generator().manual_seed(0)
            self.register_buffer('w', torch.randn(3, 10, generator=g))

        def forward(self, x, targets=None):
            feat = self.backbone(x)['concat']
            return {'stage_02': {'betas': feat @ self.w}, 'features': feat}
    return _Net().eval()


def config4_meshes(n, seed=0):
    """v = s * sum_i w_i v_i, w ~ Dirichlet(1,1,1,1), s ~ U(0.9, 1.1)
    over the 4 real v_shaped meshes; mesh 0 is the shipped sample itself."""
    import numpy as np
    from shapy_amd.utils import synthetic as syn
    faces, meshes = syn.load_topology()
    r = syn.rng_for(seed, 'config4')
    w = r.dirichlet(np.ones(4), size=n).astype(np.float32)
    s = r.uniform(0.9, 1.1, size=n).astype(np.float32)
    w[0] = [1, 0, 0, 0]
    s[0] = 1
    v = (np.einsum('nk,kvc->nvc', w, meshes) * s[:, None, None]).astype(np.float32)
    return faces, v


# ------------------------------------------------------------------------------------------
def run_measurements(args, rank, world):
    """BASELINE configs[3]: virtual measurements of `--meshes` SMPL-X meshes (per GPU)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from shapy_amd.measurements import BodyMeasurements
    data = osp.join(ROOT, 'shapy_amd', 'data')
    bm = BodyMeasurements({'meas_definition_path': f'{data}/measurement_defitions.yaml',
                           'meas_vertices_path': f'{data}/smplx_measurements.yaml',
                           'max_collisions': 256}).cuda()
    faces_np, v_np = config4_meshes(args.meshes, seed=rank)
    v = torch.from_numpy(v_np).cuda()
    f = torch.from_numpy(faces_np).cuda()
    for _ in range(args.warmup):
        out = bm.forward_vertices(v, f)
    ev0, ev1 = hip_events(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        out = bm.forward_vertices(v, f)
        ev1[i].record()
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=control_device())
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    nbytes = MEASURE_BYTES_PER_MESH * args.meshes
    achieved = nbytes / (ms * 1e-3) / 1e9
    if rank != 0:
        return None
    res = {
        'metric': 'meshes/sec, virtual measurements (mass, height, chest, waist, hips) of SMPL-X '
                  'meshes; values vs CPU oracle',
        'value': world * args.meshes * args.steps / dt, 'unit': 'meshes/sec', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'virtual measurements of {args.meshes} SMPL-X meshes per GPU '
                               '(BASELINE configs[3]: convex mixtures of the 4 shipped v_shaped '
                               'meshes, 10,475 vertices / 20,908 faces each), max_collisions 256',
                   'meshes_per_gpu': args.meshes, 'parallelism': f'dp{world}'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                     'frac_of_achievable_6290': achieved / 6290.0,
                     'kernel': 'measure_scan2_kernel + measure_hull2_kernel (one launch '
                               'group = both kernels; the time is the whole group)',
                     'bytes_per_launch_group': nbytes, 'ms_per_launch_group': ms},
    }
    if not args.no_cpu_baseline and world == 1:
        import __graft_entry__ as ge
        from oracle import measure as om
        lm = ge.oracle_state(0)['lm']
        n = min(args.meshes, 40)
        om.body_measurements(v_np[:1][:, faces_np], lm)
        t0 = time.perf_counter()
        ref = om.body_measurements(v_np[:n][:, faces_np], lm)
        dtc = time.perf_counter() - t0
        got = out[:n].cpu().numpy()
        res['cpu_baseline'] = {
            'value': n / dtc, 'unit': 'meshes/sec', 'cores': 1, 'kind': 'port',
            'cpu_model': cpu_model(),
            'sample': f'{n} meshes through the CPU oracle (C brute-force intersection + scipy '
                      f'ConvexHull, 1 thread) in {dtc:.1f} s'}
        res['parity'] = {'n_meshes': n, 'reference': 'CPU oracle', 'maxabs': {
            k: float(np.abs(got[:, i] - ref[k]).max())
            for i, k in enumerate(('mass', 'height', 'chest', 'waist', 'hips'))}}
    return res


def run_bvh(args, rank, world):
    """The LBVH path of the intersection operator (csrc/bvh.hip: build / wave-cooperative
    traverse / sort-hits), which BASELINE configs[3] itself does not exercise (its 2-triangle plane
    queries take the scan path): body-vs-body queries -- `--query-faces` triangles of the NEXT
    config-4 body against all 20,908 triangles of this one, `--meshes` pairs per GPU.
    Counterpart of mesh_mesh_intersect_cuda_op.cu:823-967 (build), :520-609 (traverse)."""
    import numpy as np
    import torch
    import mesh_mesh_intersect_cuda as mmi
    faces_np, v_np = config4_meshes(args.meshes, seed=rank)
    Q, mc = args.query_faces, args.max_collisions
    tris = torch.from_numpy(v_np).cuda()[:, torch.from_numpy(faces_np).cuda().long()].contiguous()
    q0 = 3000
    query = torch.roll(tris, -1, 0)[:, q0:q0 + Q].contiguous()       # the next body's triangles
    B, F = tris.shape[:2]
    for _ in range(args.warmup):
        f, b = mmi.mesh_to_mesh_forward(query, tris, max_collisions=mc)
    ev0, ev1 = hip_events(args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        f, b = mmi.mesh_to_mesh_forward(query, tris, max_collisions=mc)
        ev1[i].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(b_) for a, b_ in zip(ev0, ev1)]))
    overflow = int(mmi.mesh_to_mesh_forward.last_overflow.item())
    hits = int((f >= 0).sum().item())
    in_bytes = (F + Q) * 36
    out_bytes = Q * mc * (8 + 24)
    nbytes = (in_bytes + out_bytes) * B
    achieved = nbytes / (ms * 1e-3) / 1e9
    res = {
        'metric': 'mesh pairs/sec, mesh_to_mesh_forward through the LBVH (build + traverse + '
                  'sort hits); faces / barycentrics bit-exact vs CPU oracle',
        'value': B * args.steps / dt, 'unit': 'mesh pairs/sec', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'LBVH intersection: {Q} query triangles (faces {q0}..{q0 + Q} of the '
                               f'next body) vs {F} target triangles, {B} config-4 body pairs, '
                               f'max_collisions {mc}', 'pairs': B, 'query_faces': Q,
                   'max_collisions': mc},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                     'kernel': 'bvh_build_kernel + bvh_traverse_kernel + bvh_sort_hits_kernel (one '
                               'launch group; the time is the whole group incl. the output memsets)',
                     'bytes_per_launch_group': nbytes, 'ms_per_launch_group': ms,
                     'algorithmic_bytes_per_pair': {
                         'target_triangles_read': F * 36, 'query_triangles_read': Q * 36,
                         'faces_int64_and_bcs_written': out_bytes,
                         'note': 'operator contract: [B,F,3,3] + [B,Q,3,3] in, [B,Q*mc] int64 + '
                                 '[B,Q*mc,2,3] f32 out; the tree itself (keys, nodes, boxes: ~2.5 '
                                 'MB per mesh, SURVEY.md 8d) is internal traffic, not counted'}},
        'hits_per_pair': hits / B, 'overflowed_hits': overflow,
    }
    if not args.no_cpu_baseline:
        from oracle import measure as om
        n = min(B, 3)
        qn, tn = query[:n].cpu().numpy(), tris[:n].cpu().numpy()
        t0 = time.perf_counter()
        f_ref, b_ref = om.mesh_to_mesh_forward(qn, tn, mc)
        dtc = time.perf_counter() - t0
        res['cpu_baseline'] = {'value': n / dtc, 'unit': 'mesh pairs/sec', 'cores': 1, 'kind': 'port',
                               'cpu_model': cpu_model(),
                               'sample': f'{n} pairs through the C oracle (brute force, {Q} x {F} '
                                         f'SAT tests each, 1 thread) in {dtc:.1f} s'}
        res['parity'] = {'n_pairs': n, 'reference': 'C oracle (oracle/mesh_intersect.c)',
                         'faces_equal': bool(np.array_equal(f[:n].cpu().numpy(), f_ref)),
                         'bcs_equal': bool(np.array_equal(b[:n].cpu().numpy(), b_ref)),
                         'oracle_dropped': int(om.mesh_to_mesh_forward.last_dropped)}
    return res


def run_smplx(args, rank, world, net=None):
    """The SMPL-X layer alone (BASELINE configs[0] shape at --batch bodies): blend shapes,
    joint regression, pose chain, skinning, landmarks.  HBM roofline per SURVEY.md 8(d)."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    from shapy_amd.models.common.pose_utils import ContinuousRotReprDecoder
    if net is None:
        net, _ = ge.make_network()
    B = args.batch
    g = torch.Generator().manual_seed(0)
    betas = torch.randn(B, 10, generator=g).cuda()
    expr = (0.5 * torch.randn(B, 10, generator=g)).cuda()
    p6 = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(B, 22) + 0.3 * torch.randn(B, 132, generator=g)
    rot = ContinuousRotReprDecoder(22).cuda()(p6.cuda())

    def step():
        with torch.no_grad():
            return net.model(global_rot=rot[:, :1], body_pose=rot[:, 1:], betas=betas,
                             expression=expr, get_skin=True, return_shaped=True)
    for _ in range(args.warmup):
        step()
    ev0, ev1 = hip_events(args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        out = step()
        ev1[i].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    nbytes = SMPLX_CONST_BYTES + SMPLX_BYTES_PER_BODY * B
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {
        'metric': 'bodies/sec, SMPL-X forward (betas/pose -> 10,475 vertices + 123 joints)',
        'value': B * args.steps / dt, 'unit': 'bodies/sec', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'SMPL-X layer forward, batch {B} (BASELINE configs[0] shape), '
                               'synthetic model buffers', 'batch': B},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                     'kernel': 'SMPL-X launch group (blend-shape GEMMs, pose chain, skinning, '
                               'landmarks); the time is the whole group incl. host gaps',
                     'bytes_per_launch_group': nbytes, 'ms_per_launch_group': ms,
                     'floor_us_at_6290': nbytes / 6290e9 * 1e6},
    }


def _compact(r):
    """The few fields of a full bench line that an `also` sub-record keeps."""
    c = {'metric': r['metric'], 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'],
         'steps': r['steps'], 'dtype': r['dtype'], 'workload': r['config']['workload'],
         'roofline': {k: r['roofline'][k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac')}}
    if 'parity' in r:
        c['parity'] = r['parity']
    if 'cpu_baseline' in r:
        c['cpu_baseline'] = {k: r['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind', 'sample')}
    return c


def also_records(args, net, x):
    """The OTHER BASELINE.json configurations, timed in the same process right after the headline's
    timed region (rank 0, N = 1 only) so that the driver's default run sees them too: configs[0]
    (SMPL-X layer, batch 4 and 64), configs[2]'s per-GPU shard (bf16, bs 32), configs[3] (1,000
    meshes), the LBVH path of the intersection operator, and the reference's default crop size
    (256 x 256, f32, same batch).  Each is a compact record (value, ms, roofline, parity); the full
    lines come from `--workload ...` / `--dtype bf16 --batch 32` / `--size 256`."""
    import copy
    import numpy as np
    import torch
    out = {}

    def sub(**kw):
        a = copy.copy(args)
        a._sub = True
        a.steps, a.warmup = 10, 3
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    try:
        out['configs3_measurements_1000'] = _compact(run_measurements(sub(meshes=1000), 0, 1))
    except Exception as e:                            # a sub-record must never cost the headline
        out['configs3_measurements_1000'] = {'error': repr(e)}
    for b in (4, 64):
        try:
            out[f'configs0_smplx_b{b}'] = _compact(run_smplx(sub(batch=b), 0, 1, net=net))
        except Exception as e:
            out[f'configs0_smplx_b{b}'] = {'error': repr(e)}
    try:
        out['lbvh_1000_pairs'] = _compact(run_bvh(sub(meshes=1000, steps=5, warmup=2), 0, 1))
    except Exception as e:
        out['lbvh_1000_pairs'] = {'error': repr(e)}
    # configs[2]'s per-GPU shard (bf16 storage, bs 32) and the 256 x 256 crop on the SAME network
    bb = net.backbone
    keep = bb.compute_dtype
    pipe_on = getattr(args, 'pipeline', 'on') == 'on' and not getattr(args, 'single_stream', False)
    keep_x6 = bb.x6_gemm_min_batch
    for tag, dtype, B, size, pipe, x6min in (
            ('configs2_bf16_b32_per_gpu_shard', 'bf16', 32, args.size, pipe_on, keep_x6),
            ('f32_256x256_reference_default_crop', 'f32', args.batch, 256, pipe_on, keep_x6),
            ('headline_one_forward_at_a_time', args.dtype, args.batch, args.size, False, keep_x6),
            # opt-in arithmetic (--head-gemm bf16x6): the head's wide 1x1 GEMMs on the bf16 matrix cores
            ('headline_with_head_gemms_on_bf16x6', 'f32', args.batch, args.size, pipe_on, args.batch)):
        if (tag == 'headline_one_forward_at_a_time' and not pipe_on) or \
                (tag == 'headline_with_head_gemms_on_bf16x6' and (args.dtype != 'f32' or keep_x6)):
            continue
        try:
            from shapy_amd.utils import synthetic as syn
            xs = x[:B] if size == args.size else torch.from_numpy(
                syn.synthetic_images(B, size, 100)).cuda()
            nxt = {'next_images': xs} if pipe else {}
            with torch.no_grad():
                bb.compute_dtype, bb.x6_gemm_min_batch = 'f32', keep_x6
                ref = net(xs, None) if dtype != 'f32' or x6min != keep_x6 else None
                bb.compute_dtype, bb.x6_gemm_min_batch = dtype, x6min
                for _ in range(3):
                    o = net(xs, None, **nxt)
                ev0, ev1 = hip_events(10)
                idx = {'i': 0}
                h0 = bb.register_forward_pre_hook(lambda m, a, kw=None: ev0[idx['i']].record())
                h1 = bb.register_forward_hook(lambda m, a, o_: ev1[idx['i']].record())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(10):
                    idx['i'] = i
                    o = net(xs, None, **nxt)
                    o['stage_02']['betas'].cpu() if i == 9 else None
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                h0.remove(); h1.remove()
            # pipelined: the events around the call see only the rest of a forward -> the step period
            bms = dt / 10 * 1e3 if pipe else float(np.mean([a.elapsed_time(b_) for a, b_ in zip(ev0, ev1)]))
            if dtype == 'f32':
                flop = executed_mfma_flop_per_image(net, size, batch=B)
                peak = F32_MFMA_PEAK_TFLOPS
            else:
                flop = conv_flop_per_image(net, size)
                peak = BF16_MFMA_PEAK_TFLOPS
            ach = flop * B / (bms * 1e-3) / 1e12
            rec = {'metric': baseline_metric(), 'value': B * 10 / dt, 'unit': 'images/sec',
                   'ms_per_step': dt / 10 * 1e3, 'steps': 10, 'dtype': dtype, 'pipelined_batches': bool(pipe),
                   'workload': f'HRNet-W48 + regressor + SMPL-X + measurements, {size}x{size}, bs={B}, {dtype}' + (
                       ', head 1x1 GEMMs on the bf16x6 kernel' if x6min != keep_x6 else ''),
                   'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                                'frac': ach / peak, 'ms_per_launch_group': bms}}
            if ref is not None:
                rec['parity'] = {
                    'reference': 'the float32 HIP forward of the same images (the f32 path is the parity path; this '
                                 'is the arithmetic mode\'s own error on top of it)',
                    'features_maxabs': float((o['features'].float() - ref['features']).abs().max()),
                    'betas_maxabs': float((o['stage_02']['betas'].float() -
                                           ref['stage_02']['betas']).abs().max())}
            out[tag] = rec
        except Exception as e:
            out[tag] = {'error': repr(e)}
        finally:
            bb.compute_dtype, bb.x6_gemm_min_batch = keep, keep_x6
    return out


def run_regressor(args, rank, world, local_rank):
    """Returns (json dict or None, finish): `finish(res)` adds the rank-0 CPU oracle fields
    (cpu_baseline, parity) and is called by main() AFTER the process group is torn down, so no
    rank sits in a collective while rank 0 spends ~15 s on the host."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from shapy_amd.utils import synthetic as syn
    from shapy_amd import parallel

    stub = STUB['on']
    dev = 'cpu' if stub else 'cuda'
    if stub:
        net = make_stub_network(args.size)
    else:
        net, _ = ge.make_network(model_folder=f'/tmp/shapy_synth_models_r{local_rank}' if world > 1
                                 else '/tmp/shapy_synth_models')
        net.backbone.multi_stream = not args.single_stream
        net.backbone.compute_dtype = args.dtype
        if getattr(args, 'head_gemm', 'f32') == 'bf16x6':
            net.backbone.x6_gemm_min_batch = args.batch
        net.backbone.use_graph = {'auto': 'auto', 'on': True, 'off': False}[args.graph]
        if args.algo:
            net.backbone.conv_algo = args.algo
        if args.wino4_min_hw:
            net.backbone.wino4_min_hw = args.wino4_min_hw
        if args.tile_flags:
            net.backbone.tile_flags = int(args.tile_flags, 0)
        if getattr(args, 'group_branches', None):
            net.backbone.group_branches = {'auto': 'auto', 'on': True, 'off': False}[args.group_branches]
    B = args.batch
    # distinct synthetic images per rank (global batch = world * B), resident in HBM
    x_np = syn.synthetic_images(B, args.size, 100 + rank)
    x = torch.from_numpy(x_np).to(dev)
    force_gather = bool(getattr(args, 'force_gather', False)) and world == 1 and not stub
    gather_mode = getattr(args, 'gather_mode', None)
    if force_gather and gather_mode == 'work':
        # one-GPU rehearsal of the c10d fallback of the N-rank step: a world-size-1 RCCL process group,
        # so the collective, c10d's RCCL stream and the deferred join all exist in THIS process
        # (the default mode 'lane' needs no process group for one rank: shapy_amd/rccl.py)
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', init_method='env://')
    gatherer = parallel.BetasGatherer(world, force=force_gather, mode=gather_mode)
    # the timed region includes the D2H of the betas (async copy into pinned host
    # memory on the compute stream; the closing synchronize covers the last one)
    betas_host = torch.empty(B, 10, dtype=torch.float32)
    if not stub:
        betas_host = betas_host.pin_memory()

    # software pipelining of consecutive batches (shapy_amd/models/backbone/prefetch.py): every step hands the
    # NEXT batch (here: the same resident tensor) to the network, whose stem + layer1 then run under this
    # batch's stage 4 / head / SMPL-X tail.  The last timed step prefetches too: the timed region holds exactly
    # `steps` prologues and `steps` rests (the first step's prologue ran in the last warmup step, before the
    # synchronize in front of the clock).  Outputs are bit-identical with --pipeline off.
    pipelined = getattr(args, 'pipeline', 'on') == 'on' and not stub and not getattr(args, 'single_stream', False)
    nxt = {'next_images': x} if pipelined else {}

    def step():
        with torch.no_grad():
            out = net(x, None, **nxt)
            betas_host.copy_(out['stage_02']['betas'], non_blocking=True)
            betas = gatherer(out['stage_02']['betas'])     # joined at the next call / wait()
        return out, betas

    for _ in range(args.warmup):
        step()
    gatherer.wait()
    ev0, ev1 = hip_events(args.steps)
    # events around the backbone call are recorded on the launch stream by a forward hook pair
    idx = {'i': 0}
    h0 = net.backbone.register_forward_pre_hook(lambda m, a: ev0[idx['i']].record())
    h1 = net.backbone.register_forward_hook(lambda m, a, o: ev1[idx['i']].record())

    device_sync()
    if world > 1:
        barrier()
    device_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        idx['i'] = i
        out, betas = step()
    gatherer.wait()
    device_sync()
    own_dt = time.perf_counter() - t0
    if world > 1:
        barrier()
    device_sync()
    dt = time.perf_counter() - t0
    h0.remove(); h1.remove()
    per_rank = None
    if world > 1:
        cdev = control_device()
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        own = torch.tensor([B * args.steps / own_dt], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allr, own)
        per_rank = [float(t.item()) for t in allr]
    assert betas.shape == (world * B, 10)
    betas_all = betas.clone()
    # after the clock: how long does the lane-1 gather keep the NEXT step's join waiting?  Three extra steps, each
    # joined from the host right after the following step has been enqueued (BetasGatherer.wait(measure=True)); at
    # N > 1 a slow rank would show up here as a lane-1 stall -- 0 when the gather finished long before.
    join_wait_ms = None
    if (world > 1 or force_gather) and not stub and gatherer.mode == 'lane':
        waits = []
        for _ in range(3):
            gatherer.wait()
            with torch.no_grad():
                o2 = net(x, None)
                gatherer(o2['stage_02']['betas'])
                net.backbone(x)                      # the next step's backbone is in the queues ...
            gatherer.wait(measure=True)              # ... when the host asks for the gather
            waits.append(gatherer.last_join_wait_ms)
        device_sync()
        join_wait_ms = float(max(waits))
    gatherer.close()                   # (the communicator goes before the process group does)
    betas = betas_all
    if force_gather:
        assert torch.equal(betas, out['stage_02']['betas']) and gatherer.issued >= args.steps + args.warmup
        if dist.is_initialized():
            dist.destroy_process_group()
    assert torch.equal(betas_host, out['stage_02']['betas'].cpu())      # the D2H copy landed
    if world > 1:      # the gathered tensor really holds every rank's betas: own shard in place
        assert torch.equal(betas[rank * B:(rank + 1) * B], out['stage_02']['betas'])
        # ... and every OTHER rank's shard is that rank's own result (checked on rank 0 against
        # the betas each rank sends separately)
        mine = out['stage_02']['betas'].contiguous().to(control_device())
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        assert torch.equal(betas.to(mine.device), torch.cat(every, dim=0))

    if stub:
        if rank != 0:
            return None, None
        res = {'metric': baseline_metric(), 'value': world * B * args.steps / dt, 'unit': 'images/sec',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'stub': True,
               'config': {'workload': 'CPU STUB (--cpu-stub): control-flow test of the N-rank path on '
                                      'gloo, NOT a measurement', 'global_batch': world * B,
                          'parallelism': f'dp{world}'},
               'rccl_ranks': world, 'backend': 'gloo',
               'per_rank': {'images_per_sec': per_rank,
                            'allgather': {'issued': gatherer.issued,
                                          'joined_by_next_step': gatherer.deferred_waits}}}
        return res, None

    backbone_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    call_ms = backbone_ms
    if pipelined and net.backbone._prefetch.used:
        # the events around the call bracket only the REST of a forward (its stem + layer1 ran on a side stream
        # under the previous step): the roofline takes the whole step period instead -- an upper bound of
        # the time the chip spends per backbone forward (it includes the regressor / SMPL-X tail)
        backbone_ms = own_dt / args.steps * 1e3
    flop_img = conv_flop_per_image(net, args.size)
    algorithmic = flop_img * B / (backbone_ms * 1e-3) / 1e12
    # `achieved` = FLOPs the matrix cores EXECUTE / backbone time, so that frac <= 1 by
    # construction: the Winograd layers execute 2.25x (F(2x2)) / 4x (F(4x4)) fewer multiplies than
    # the direct convolution they compute.  The direct-convolution-equivalent rate (SURVEY.md 8d:
    # 36.933 GFLOP per image) is reported next to it as `algorithmic_equiv_tflops`.
    exec_img = executed_mfma_flop_per_image(net, args.size, batch=B) if args.dtype == 'f32' else flop_img
    x6_parts = executed_mfma_flop_per_image(net, args.size, batch=B, parts=True) if args.dtype == 'f32' else None
    achieved = exec_img * B / (backbone_ms * 1e-3) / 1e12
    # f32x6 issues 6 bf16 MFMAs per float32 multiply-add: its matrix-core roof in algorithmic
    # (float32) FLOP/s is the dense bf16 peak / 6
    peak = {'f32': F32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS,
            'f32x6': BF16_MFMA_PEAK_TFLOPS / 6.0}[args.dtype]
    kernel = {'f32': 'conv_igemm_kernel<F32> (v_mfma_f32_16x16x4_f32)',
              'bf16': 'conv_igemm_kernel<BF16> (v_mfma_f32_16x16x32_bf16)',
              'f32x6': 'conv_x6_kernel (6 x v_mfma_f32_16x16x32_bf16 per f32 product; peak = '
                       'dense bf16 peak / 6)'}[args.dtype]
    algo = getattr(net.backbone, 'conv_algo', 'direct')
    if args.dtype == 'f32' and algo == 'winograd4':
        kernel += f' + conv_wino4_kernel (Winograd F(4x4,3x3), maps >= {net.backbone.wino4_min_hw} ' \
                  'px) + conv_wino_kernel (F(2x2,3x3), the other 3x3 stride-1 layers)'
    elif args.dtype == 'f32' and algo != 'direct':
        kernel += f' + conv_wino_kernel (Winograd F(2x2,3x3) for the 3x3 stride-1 layers, ' \
                  f'algo={algo})'
    # HBM bytes per launch group (one backbone forward) from the committed PMC passes, gfx950
    # FETCH x2 correction applied
    traffic = pmc_traffic(B, args.size, args.dtype, algo)
    n_launch = launches_per_forward(_f32_plan(net, args.size, B) if args.dtype == 'f32' else
                                    next(iter(net.backbone._engine.values()))['plan'])
    traffic_other = None
    if traffic is None:      # no PMC pass of THIS algorithm yet: `traffic` stays null; the last
        # measured build of the same workload is quoted as context, labelled with its algorithm
        traffic_other = pmc_traffic(B, args.size, args.dtype, algo, any_algo=True)
    if rank != 0:
        return None, None
    res = {
        'metric': baseline_metric(),
        'value': world * B * args.steps / dt,
        'unit': 'images/sec',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': f'HRNet-W48 + iterative regressor + SMPL-X + virtual '
                               f'measurements, random-init weights, {args.size}x{args.size}, '
                               f'bs={B} per GPU, ' + {
                                   'f32': 'fp32 (BASELINE configs[1])',
                                   'f32x6': 'fp32 tensors, bf16x6 split products '
                                            '(BASELINE configs[1])',
                                   'bf16': 'bf16 storage / f32 accumulate (BASELINE '
                                           'configs[2] precision)'}[args.dtype],
                   'global_batch': world * B, 'parallelism': f'dp{world}',
                   'multi_stream': not args.single_stream, 'conv_algo': algo,
                   'd2h_betas_in_timed_region': True,
                   'pipelined_batches': bool(pipelined and net.backbone._prefetch.used),
                   'head_gemm_arithmetic': ('bf16x6 (float32 tensors, exact 3-way bf16 split, f32 accumulate)'
                                            if x6_parts and x6_parts[1] else 'f32 MFMA'),
                   'wino4_ksplit': {f'{c}@{t}': sl for (c, t), sl in net.backbone.ksplit_policy(B).items()},
                   'hip_graph': bool(net.backbone.use_graph is True or
                                     (net.backbone.use_graph == 'auto' and
                                      B <= net.backbone.graph_max_batch))},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak,
                     'unit': 'TFLOP/s', 'frac': achieved / peak,
                     'traffic': traffic['bytes_fetch_x2_corrected'] if traffic else None,
                     'traffic_detail': traffic if traffic or not traffic_other else {
                         'note': f'no rocprofv3 --pmc pass of conv_algo={algo} yet (traffic: null); '
                                 'last measured build of the same workload, for context only',
                         'other_build': traffic_other},
                     'peak_microbenchmark': {'f32': 155.0, 'bf16': 2495.0,
                                             'f32x6': 2495.0 / 6.0}[args.dtype],   # MI355X_MICROARCH.md
                     'kernel': kernel + f', {n_launch} launches per backbone forward',
                     'flop_per_launch_group': exec_img * B,
                     'bf16x6_layers': None if not (x6_parts and x6_parts[1]) else {
                         'f32_core_flop_per_launch_group': x6_parts[0] * B,
                         'bf16_core_flop_per_launch_group': x6_parts[1] * B,
                         'note': 'the head\'s wide 1x1 GEMMs (Cin, Cout >= 512) run on the bf16 matrix cores: float32 '
                                 'tensors, products from the exact 3-way bf16 split (6 MFMAs per product, f32 '
                                 'accumulate; same kernel-test tolerance as the f32 kernel).  In `achieved` they '
                                 f'count with the matrix-pipe time they need: bf16 FLOPs x f32 peak / bf16 peak '
                                 f'(= {X6_F32_PIPE_EQUIV:.4f} per float32 FLOP)'},
                     'ms_per_launch_group': backbone_ms,
                     # north_star's "MFMA utilisation": matrix-pipe busy cycles of one forward from the counter
                     # (SQ_VALU_MFMA_BUSY_CYCLES, PMC pass over the four-lane plan: traffic_detail.pmc_plan) over the
                     # SIMD cycles of the MEASURED step period of THIS run -- at the nominal 2.4 GHz (a lower bound:
                     # these boxes hold ~2.06-2.1 GHz under MFMA load) and at 2.1 GHz
                     'mfma_busy': None if not (traffic and traffic.get('mfma_busy_cycles_per_forward')) else {
                         'busy_cycles_per_forward': traffic['mfma_busy_cycles_per_forward'],
                         'frac_of_simd_cycles_at_2.4GHz': traffic['mfma_busy_cycles_per_forward'] /
                                                         (1024 * 2.4e9 * backbone_ms * 1e-3),
                         'frac_of_simd_cycles_at_2.1GHz': traffic['mfma_busy_cycles_per_forward'] /
                                                         (1024 * 2.1e9 * backbone_ms * 1e-3),
                         'source': traffic['source'], 'pmc_plan': traffic.get('pmc_plan')},
                     'duration': ('step period (pipelined: the next batch\'s stem + layer1 run under this batch\'s '
                                  f'head on a side stream; HIP events around the call see only the rest: {call_ms:.3f} ms)'
                                  if backbone_ms != call_ms else 'HIP events around the backbone call'),
                     'achieved_counts': 'FLOPs issued to the matrix cores (direct layers: 2 x MAC; '
                                        'Winograd layers: 16 products per 2x2 tile / 36 per 4x4 tile '
                                        'and channel pair, whole tiles) / backbone time (HIP events)',
                     'algorithmic_equiv_tflops': algorithmic,
                     'algorithmic_equiv_frac_of_peak': algorithmic / peak,
                     'algorithmic_flop_per_launch_group': flop_img * B},
    }
    if world > 1:
        res['rccl_ranks'] = world
        res['per_rank'] = {'images_per_sec': per_rank,
                           'allgather': {'issued': gatherer.issued, 'mode': gatherer.mode,
                                         'joined_by_next_step': gatherer.deferred_waits,
                                         'lane_join_wait_ms_max_after_clock': join_wait_ms}}
    if force_gather:
        res['rccl_ranks'] = 1
        res['force_gather'] = {'mode': gatherer.mode, 'issued': gatherer.issued,
                               'joined_by_next_step': gatherer.deferred_waits,
                               'lane_join_wait_ms_max_after_clock': join_wait_ms,
                               'note': 'world-size-1 RCCL communicator: the all_gather of the N-rank step on '
                                       'ONE GPU (work: through c10d, with its RCCL stream)'}
    import hashlib
    res['betas_sha1'] = hashlib.sha1(betas_host.numpy().tobytes()).hexdigest()[:16]
    if world == 1 and not getattr(args, 'no_also', False) and not getattr(args, '_sub', False):
        res['also'] = also_records(args, net, x)
        # the same values as flat scalars (a line parser that keeps only top-level scalars keeps these)
        for tag, rec in res['also'].items():
            if isinstance(rec, dict) and 'value' in rec:
                res[f'also_{tag}_value'] = rec['value']
                res[f'also_{tag}_ms'] = rec['ms_per_step']
                res[f'also_{tag}_roofline_frac'] = rec['roofline']['frac']

    def finish(res):                                 # rank 0 only; after destroy_process_group()
        if not args.no_cpu_baseline:
            base, par = cpu_baseline_and_parity(x_np, out, args.size)
            res['parity'] = par
            if world == 1:
                res['cpu_baseline'] = base
        return res
    return res, finish


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='regressor', choices=['regressor', 'measurements', 'smplx', 'bvh'],
                    help='regressor = BASELINE configs[1] (headline); measurements = configs[3]; '
                         'smplx = the SMPL-X layer alone (configs[0] shape)')
    ap.add_argument('--batch', type=int, default=64, help='images (bodies) per GPU')
    ap.add_argument('--meshes', type=int, default=1000, help='meshes per GPU (measurements)')
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--query-faces', type=int, default=700, help='bvh: query triangles per pair')
    ap.add_argument('--max-collisions', type=int, default=64, help='bvh: hits kept per query triangle')
    ap.add_argument('--no-cpu-baseline', action='store_true',
                    help='skip the CPU oracle (cpu_baseline and parity fields)')
    ap.add_argument('--single-stream', action='store_true')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the backbone as one hipGraph (auto: never since round 3 -- the eager '
                         'event-driven forward is faster at every batch size; on: the captured barrier plan)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f32x6', 'bf16'],
                    help='f32 = BASELINE configs[1] (headline) on the f32 matrix-core path; '
                         'f32x6 = same float32 tensors, products from the exact 3-way bf16 split '
                         '(6 bf16 MFMAs per product); bf16 = configs[2] storage type')
    ap.add_argument('--algo', default=None, choices=['direct', 'auto', 'winograd', 'winograd4'],
                    help='f32 conv algorithm override (default: the backbone\'s own default)')
    ap.add_argument('--tile-flags', default='',
                    help='A/B knob bits OR-ed into every conv\'s tile id (shapy_amd/_lib.py), e.g. '
                         '0x10000 = m-major XCD order for large weights')
    ap.add_argument('--cpu-stub', action='store_true',
                    help='TEST HARNESS, never a measurement: gloo ranks + a stub CPU forward, to run the '
                         'N-rank control flow of this file without GPUs (tests/test_host_cpu.py); '
                         'the line it prints carries "stub": true')
    ap.add_argument('--wino4-min-hw', type=int, default=0,
                    help='--algo winograd4: smallest map side that takes F(4x4,3x3) (default: the '
                         'backbone\'s own, 7)')
    ap.add_argument('--force-gather', action='store_true',
                    help='N = 1 only: create a world-size-1 RCCL group and take BetasGatherer\'s '
                         'collective path (rehearsal of the N-rank step on one GPU; the line says '
                         '"rccl_ranks": 1)')
    ap.add_argument('--gather-mode', default=None, choices=['lane', 'work'],
                    help='BetasGatherer issue mode (shapy_amd/parallel.py; default: lane = ncclAllGather '
                         'called directly on the executor\'s lane-1 stream, joined one step later)')
    ap.add_argument('--control-backend', default='gloo', choices=['gloo', 'nccl'],
                    help='torch.distributed backend of the control plane at N > 1 (barriers, timing '
                         'reduction, RCCL id exchange); the betas all-gather is RCCL either way')
    ap.add_argument('--group-branches', default=None, choices=['auto', 'on', 'off'],
                    help='persistent grouped F(4x4) launches per depth level of a module '
                         '(HighResolutionNet.group_branches)')
    ap.add_argument('--pipeline', default='on', choices=['on', 'off'],
                    help="on (default): every step passes the next batch to the network (next_images=), whose stem + "
                         "layer1 run under the current batch's head; off: one forward at a time")
    ap.add_argument('--head-gemm', default='f32', choices=['f32', 'bf16x6'],
                    help='bf16x6: the head\'s wide 1x1 GEMMs (Cin, Cout >= 512) on the bf16 matrix cores -- float32 '
                         'tensors, exact 3-way bf16 split, f32 accumulate (opt-in; the default run reports it as an '
                         '`also` record)')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the `also` sub-records (the other BASELINE configurations, timed after '
                         'the headline\'s timed region in the default N = 1 run)')
    args = ap.parse_args()

    STUB['on'] = bool(args.cpu_stub)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_spawn(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'--nproc-per-node {args.gpus} (or without a launcher: bench.py spawns '
                         'its own ranks)')
    if STUB['on'] and args.workload != 'regressor':
        raise SystemExit('--cpu-stub only exercises the regressor workload\'s rank plumbing')
    if not STUB['on']:
        torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # explicit timeout: a rank that dies must fail the job in minutes, not hang the node
        # control plane (rendezvous, barriers, max-over-ranks of the timing, the 128-byte RCCL id):
        # gloo by default -- the DATA plane, the all-gather of the betas, is RCCL called directly on
        # the compute stream (shapy_amd/rccl.py); c10d's NCCL backend would add a stream per process,
        # which alone costs the four-lane backbone 17 % (profiles/r04j_*)
        backend = 'gloo' if STUB['on'] else args.control_backend
        dist.init_process_group(backend, init_method='env://', timeout=datetime.timedelta(seconds=600))
    if (rank == 0 and not STUB['on']
            and not osp.exists(osp.join(ROOT, 'shapy_amd', 'csrc', 'libshapy_hip.so'))):
        from shapy_amd import build as hip_build      # fresh checkout: the library is git-ignored
        hip_build.build()
    if world > 1:
        barrier()

    finish = None
    if args.workload == 'measurements':
        res = run_measurements(args, rank, world)
    elif args.workload == 'smplx':
        res = run_smplx(args, rank, world)
    elif args.workload == 'bvh':
        res = run_bvh(args, rank, world)
    else:
        res, finish = run_regressor(args, rank, world, local_rank)
    if world > 1:
        # every collective of the job is behind us: tear the group down BEFORE rank 0 turns to the
        # CPU oracle, so that ranks 1..N-1 exit instead of waiting in a barrier for ~15 s
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        if finish is not None:
            res = finish(res)
        # the JSON line is the LAST thing on stdout: whatever native libraries (RCCL prints its library
        # path through C stdio) still hold in their buffers goes out first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
