"""Headline benchmark: images/sec of the SHAPY hot path (HRNet-W48 + iterative regressor +
SMPL-X + virtual measurements) on synthetic 224x224 crops, batch 64 per GPU, float32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full forward of the regressor on one batch that is already resident in HBM
(BASELINE.json configs[1]: "HRNet-W48 + SMPL-X head, random-init weights, 224x224 bs=64 fp32
on 1xMI355X"); with N > 1 every rank runs its own shard (weak scaling: 64 images per GPU) and
the predicted betas are all-gathered with RCCL at the end of every step.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      the MFMA roofline of the dominant kernel family (conv_igemm_f32, 330 launches
                per backbone forward): algorithmic conv FLOPs (2 x 18,466,524,160 MAC per image,
                SURVEY.md 8d) / time of the backbone call, measured with HIP events on the
                launch stream inside the timed loop; peak = 157.3 TFLOP/s (f32 MFMA, dense)
  cpu_baseline  the CPU oracle (torch CPU restatement of the reference, "port") timed on this
                host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import os.path as osp
import sys
import time

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

CONV_FLOP_PER_IMAGE_224 = 2 * 18_466_524_160       # SURVEY.md 8(d), counted from the reference
F32_MFMA_PEAK_TFLOPS = 157.3                       # MI355X_MICROARCH.md, dense f32 MFMA
BF16_MFMA_PEAK_TFLOPS = 2500.0                     # dense bf16 MFMA (not the 2:1-sparse figure)


def baseline_metric():
    """The metric string of BASELINE.json, verbatim."""
    try:
        with open(osp.join(ROOT, 'BASELINE.json')) as fh:
            return json.load(fh)['metric']
    except (OSError, KeyError, ValueError):
        return 'images/sec whole-node (HRNet+SMPL-X fwd), 224\u00d7224 bs=64; betas L2 vs CPU'


def conv_flop_per_image(net, size):
    plan = net.backbone._build_plan(size, size)      # f32 plan: algorithmic (unpadded) MACs
    macs = sum(o['Ho'] * o['Wo'] * o['Cout'] * o['Cin'] * o['ksize'] ** 2
               for o in plan.ops if o['type'] != 2)
    return 2 * macs


def pmc_traffic(batch, size):
    """HBM bytes per backbone forward from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_hbm_traffic.json; FETCH_SIZE and WRITE_SIZE need separate passes and cannot
    be collected from inside this process).  None when the workload differs."""
    import glob
    for f in sorted(glob.glob(osp.join(ROOT, 'profiles', '*_pmc_hbm_traffic.json')), reverse=True):
        with open(f) as fh:
            d = json.load(fh)
        h = d.get('hbm_bytes_per_backbone_forward', {})
        if h.get('batch') == batch and h.get('size') == size:
            return {'bytes_as_reported': h['as_reported'], 'bytes_fetch_x2_corrected':
                    h['fetch_x2_corrected'], 'source': osp.relpath(f, ROOT)}
    return None


def cpu_baseline(batch, size, budget_s=15.0):
    """Times the CPU oracle (kind 'port') for about `budget_s` seconds of work."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    from shapy_amd.utils import synthetic as syn
    cores = min(torch.get_num_threads(), 64)      # more threads only add scheduling noise here
    torch.set_num_threads(cores)
    b = min(batch, 16)
    x = syn.synthetic_images(b, size, 1)
    state = ge.oracle_state(0)
    ge.oracle_forward(x[:1], state=state)                    # untimed warm-up pass
    n, t0 = 0, time.perf_counter()
    while True:
        ge.oracle_forward(x, state=state)
        n += b
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cpu = ''
    try:
        with open('/proc/cpuinfo') as fh:
            cpu = next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), '')
    except OSError:
        pass
    return {'value': n / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port', 'cpu_model': cpu,
            'host_logical_cpus': os.cpu_count(),
            'sample': f'{n} images ({size}x{size}, batches of {b}) through the CPU oracle '
                      f'(torch-CPU HRNet + numpy SMPL-X + C intersection + scipy hull) in {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--single-stream', action='store_true')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the backbone as one hipGraph (auto: batches <= 8)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f32x6', 'bf16'],
                    help='f32 = BASELINE configs[1] (headline) on the f32 matrix-core path; '
                         'f32x6 = same float32 tensors, products from the exact 3-way bf16 split '
                         '(6 bf16 MFMAs per product); bf16 = configs[2] storage type')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from shapy_amd.utils import synthetic as syn
    from shapy_amd import parallel

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks '
                         f'(WORLD_SIZE={world})')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', init_method='env://')
    if rank == 0 and not osp.exists(osp.join(ROOT, 'shapy_amd', 'csrc', 'libshapy_hip.so')):
        from shapy_amd import build as hip_build      # fresh checkout: the library is git-ignored
        hip_build.build()
    if world > 1:
        dist.barrier()

    net, _ = ge.make_network(model_folder=f'/tmp/shapy_synth_models_r{local_rank}' if world > 1
                             else '/tmp/shapy_synth_models')
    net.backbone.multi_stream = not args.single_stream
    net.backbone.compute_dtype = args.dtype
    net.backbone.use_graph = {'auto': 'auto', 'on': True, 'off': False}[args.graph]
    B = args.batch
    # distinct synthetic images per rank (global batch = world * B), resident in HBM
    x = torch.from_numpy(syn.synthetic_images(B, args.size, 100 + rank)).cuda()
    gatherer = parallel.BetasGatherer(world)

    def step():
        with torch.no_grad():
            out = net(x, None)
            betas = gatherer(out['stage_02']['betas'])
        return out, betas

    for _ in range(args.warmup):
        step()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    # events around the backbone call are recorded on the launch stream by a forward hook pair
    idx = {'i': 0}
    h0 = net.backbone.register_forward_pre_hook(lambda m, a: ev0[idx['i']].record())
    h1 = net.backbone.register_forward_hook(lambda m, a, o: ev1[idx['i']].record())

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        idx['i'] = i
        out, betas = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h0.remove(); h1.remove()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert betas.shape == (world * B, 10)

    backbone_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    flop_img = conv_flop_per_image(net, args.size)
    achieved = flop_img * B / (backbone_ms * 1e-3) / 1e12
    # f32x6 issues 6 bf16 MFMAs per float32 multiply-add: its matrix-core roof in algorithmic
    # (float32) FLOP/s is the dense bf16 peak / 6
    peak = {'f32': F32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS,
            'f32x6': BF16_MFMA_PEAK_TFLOPS / 6.0}[args.dtype]
    kernel = {'f32': 'conv_igemm_kernel<F32> (v_mfma_f32_16x16x4_f32)',
              'bf16': 'conv_igemm_kernel<BF16> (v_mfma_f32_16x16x32_bf16)',
              'f32x6': 'conv_x6_kernel (6 x v_mfma_f32_16x16x32_bf16 per f32 product; peak = '
                       'dense bf16 peak / 6)'}[args.dtype]
    # HBM bytes per launch group (one backbone forward) from the committed PMC passes, gfx950
    # FETCH x2 correction applied; measured on the f32 build
    traffic = pmc_traffic(B, args.size) if args.dtype == 'f32' else None
    if rank == 0:
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
                       'multi_stream': not args.single_stream,
                       'hip_graph': bool(net.backbone.use_graph is True or
                                         (net.backbone.use_graph == 'auto' and
                                          B <= net.backbone.graph_max_batch))},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic['bytes_fetch_x2_corrected'] if traffic else None,
                         'traffic_detail': traffic,
                         'peak_sustained_measured': {'f32': 141.0, 'bf16': 1410.0,
                                                     'f32x6': 1410.0 / 6.0}[args.dtype],
                         'kernel': kernel + ', 330 launches per backbone forward',
                         'flop_per_launch_group': flop_img * B,
                         'ms_per_launch_group': backbone_ms},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            res['cpu_baseline'] = cpu_baseline(B, args.size)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
