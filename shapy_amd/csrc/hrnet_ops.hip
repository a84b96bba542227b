// HRNet op-list executor + the two non-GEMM kernels of the backbone (stem conv, spatial mean).
// Reference: regressor/human_shape/models/backbone/hrnet.py:426-498.
#include <mutex>

#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace shapy {

// ---- stem: conv3x3 s2 p1, 3 -> 64 channels, NCHW input -> NHWC output, folded BN + ReLU ----
// (hrnet.py:427-429).  K = 27 is too small for the matrix cores; 0.5 % of the network's MACs.
__device__ __forceinline__ void store_elem(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_elem(unsigned short *p, float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);                      // bf16, round to nearest even
  *p = (unsigned short)(u >> 16);
}
__device__ __forceinline__ void store_row8(float *p, const float (&r)[8]) {
  reinterpret_cast<f32x4 *>(p)[0] = f32x4{r[0], r[1], r[2], r[3]};
  reinterpret_cast<f32x4 *>(p)[1] = f32x4{r[4], r[5], r[6], r[7]};
}
__device__ __forceinline__ void store_row8(unsigned short *p, const float (&r)[8]) {
  u32x4 x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned lo = __float_as_uint(r[2 * e]), hi = __float_as_uint(r[2 * e + 1]);
    lo += 0x7fffu + ((lo >> 16) & 1u);                  // bf16, round to nearest even
    hi += 0x7fffu + ((hi >> 16) & 1u);
    x[e] = (lo >> 16) | (hi & 0xffff0000u);
  }
  *reinterpret_cast<u32x4 *>(p) = x;
}
__device__ __forceinline__ float load_elem(const float *p) { return *p; }
__device__ __forceinline__ float load_elem(const unsigned short *p) {
  return __uint_as_float((unsigned)*p << 16);
}

template <typename OutT>
__global__ __launch_bounds__(256, 4) void stem_conv_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ wgt,
                                                        const float *__restrict__ bias,
                                                        OutT *__restrict__ out, int B, int H, int W,
                                                        int Ho, int Wo, int out_ld) {
  // 256 threads = 8 segments of 16 consecutive output pixels (Wo % 16 == 0: a segment never leaves
  // its row) x 4 pixel quads x 8 groups of 8 output channels: a thread owns 4 consecutive pixels x 8
  // channels.  Each segment first stages its input patch (3 channels x 3 rows x 33 columns, zero
  // padding applied) in LDS with coalesced row reads.
  // History: round 2 had every thread fetch its 27 taps from global memory (load-issue-bound, 172 us
  // at B = 64); round 3 staged the patch but a thread owned ONE pixel x 8 channels and read 36 bytes
  // of LDS per 8 FMAs -- 249 KB of LDS reads per 32 pixels, LDS-bandwidth-bound at 107 us for a 50 us
  // job by HBM traffic.  Four pixels per thread share every weight read (3.3x fewer LDS bytes per
  // FMA).  Same taps in the same order (kh, kw, c) per output -> bit-identical results.
  __shared__ __attribute__((aligned(16))) float w[27 * 64];   // w[k][n], k = (kh*3+kw)*3 + c
  __shared__ __attribute__((aligned(16))) float patch[8][3][3][36];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) {
    const int n = i & 63, k = i >> 6;
    w[i] = wgt[n * 27 + k];
  }
  const long npix = (long)B * Ho * Wo;
  {
    // staging: 32 threads per segment
    const int seg = threadIdx.x >> 5, ts = threadIdx.x & 31;
    const long seg_pix0 = (long)blockIdx.x * 128 + seg * 16;
    if (seg_pix0 < npix) {
      const int wo0 = (int)(seg_pix0 % Wo);
      const long tq = seg_pix0 / Wo;
      const int ho = (int)(tq % Ho);
      const int b = (int)(tq / Ho);
      const float *inb = in + (long)b * 3 * H * W;
      for (int idx = ts; idx < 3 * 3 * 33; idx += 32) {
        const int c = idx / 99, r = (idx % 99) / 33, j = idx % 33;
        const int hi = ho * 2 - 1 + r, wi = wo0 * 2 - 1 + j;
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        patch[seg][c][r][j] = ok ? inb[((long)c * H + hi) * W + wi] : 0.f;
      }
    }
  }
  __syncthreads();
  const int g = threadIdx.x & 7, quad = threadIdx.x >> 3;
  const int seg = quad >> 2, q = quad & 3;
  const long pix0 = (long)blockIdx.x * 128 + seg * 16 + q * 4;
  if (pix0 >= npix) return;                       // (whole segments: npix % 16 == 0)
  // (packed FMAs: v_pk_fma_f32 does two IEEE fmas per lane and instruction -- the plain form is
  // VALU-bound at 40 us for B = 64)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[4][4];
#pragma unroll
  for (int px = 0; px < 4; ++px)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[px][i] = f32x2{0.f, 0.f};
  // (kh stays a loop: fully unrolled, hipcc requests all 54 weight fragments first and spills)
#pragma unroll 1
  for (int kh = 0; kh < 3; ++kh) {
    // columns 8 q .. 8 q + 8 of the three channel rows: pixel px reads column 2 px + kw of them
    float x[3][9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float *row = &patch[seg][c][kh][8 * q];
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(row);
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(row + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[c][e] = lo[e];
        x[c][4 + e] = hi[e];
      }
      x[c][8] = row[8];
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float *wk = w + ((kh * 3 + kw) * 3 + c) * 64 + g * 8;
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wk);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wk + 4);
        const f32x2 wp[4] = {f32x2{w0[0], w0[1]}, f32x2{w0[2], w0[3]}, f32x2{w1[0], w1[1]},
                             f32x2{w1[2], w1[3]}};
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const float xs = x[c][2 * px + kw];
          const f32x2 xv = f32x2{xs, xs};
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[px][i] = __builtin_elementwise_fma(xv, wp[i], acc[px][i]);
        }
      }
    }
  }
  // the 8 channels of a thread as 16-byte stores (8 lanes = a pixel's whole 256-byte row)
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = bias[g * 8 + i];
  const bool vec = (out_ld & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    OutT *o = out + (pix0 + px) * out_ld + g * 8;
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = relu_keep_nan(acc[px][i >> 1][i & 1] + bs[i]);
    if (vec) {
      store_row8(o, r);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) store_elem(o + i, r[i]);
    }
  }
}

// ---- spatial mean over H*W (hrnet.py:484) ----
template <typename InT>
__global__ void mean_pool_kernel(const InT *__restrict__ in, float *__restrict__ out, int HW,
                                 int C, int in_ld, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long b = i / C;
  const InT *p = in + b * HW * in_ld + c;
  float s = 0.f;
  for (int k = 0; k < HW; ++k) s += load_elem(p + (long)k * in_ld);
  out[i] = s / (float)HW;
}

// ---- side streams for the independent branches of a HighResolutionModule ----
constexpr int N_SIDE = 6;        // side streams: lanes 1..3 = branches, 4..6 = auxiliary chains
constexpr int N_EVENTS = 64;     // dependency events (ShapyOp.sig / .wait), reused from epoch to epoch
struct Lanes {
  hipStream_t s[N_SIDE] = {};
  hipEvent_t fork = nullptr;
  hipEvent_t join[N_SIDE] = {};
  hipEvent_t ev[N_EVENTS] = {};
  bool ready = false;
};
static Lanes g_lanes[16];
static std::mutex g_lanes_mu;
static std::mutex g_run_mu;      // held while a forward is enqueued (and by anybody else who touches the lanes)

// Side stream of lane li + 1, created on first use: HIP spreads streams over a few hardware queues,
// so streams that no plan uses should not exist.  Lane i + 1 (the branch with the smaller maps = the
// longer chain of latency-bound launches) gets a higher stream priority than lane i (+1.1..1.6 % end
// to end, run T of round 3).
static int lane_stream(Lanes *L, int li, hipStream_t *out) {
  if (!L->s[li]) {
    int plo = 0, phi = 0;
    SHAPY_HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));    // phi = highest (numerically lowest)
    // (the device has three levels, lowest 1 .. highest -1: lane 1 gets 0, lanes 2 and 3 the highest; seven other
    // assignments measured within +-0.7 % on the round-6 kernels, profiles/r06g_lane_priorities_ab.txt)
    int pr = plo - (li % 3 + 1);
    if (pr < phi) pr = phi;
    SHAPY_HIP_TRY(hipStreamCreateWithPriority(&L->s[li], hipStreamNonBlocking, pr));
  }
  *out = L->s[li];
  return SHAPY_OK;
}

static int get_lanes(Lanes **out) {
  int dev = 0;
  SHAPY_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return SHAPY_EINVAL;
  std::lock_guard<std::mutex> lk(g_lanes_mu);
  Lanes &L = g_lanes[dev];
  if (!L.ready) {
    for (int i = 0; i < N_SIDE; ++i)
      SHAPY_HIP_TRY(hipEventCreateWithFlags(&L.join[i], hipEventDisableTiming));
    SHAPY_HIP_TRY(hipEventCreateWithFlags(&L.fork, hipEventDisableTiming));
    for (int i = 0; i < N_EVENTS; ++i)
      SHAPY_HIP_TRY(hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming));
    L.ready = true;
  }
  *out = &L;
  return SHAPY_OK;
}

// The side stream of `lane` (1 .. N_SIDE) on the current device, created on first use: lets the host
// put work of its own -- the betas all-gather of a data-parallel step -- on a stream the process
// already has instead of adding one (every extra stream costs the four-lane forward).
int hrnet_lane_stream(int lane, hipStream_t *out) {
  if (lane < 1 || lane > N_SIDE || !out) return SHAPY_EINVAL;
  Lanes *L = nullptr;
  std::lock_guard<std::mutex> lk(g_run_mu);
  const int rc = get_lanes(&L);
  if (rc) return rc;
  return lane_stream(L, lane - 1, out);
}

int hrnet_run(const ShapyOp *ops, int n_ops, const void *weights, const float *input, void *ws,
              int64_t ws_per_img, int32_t *counters, int64_t cnt_per_img, float *features_out, int B,
              int H, int W, int multi_stream, int dtype, hipStream_t main) {
  const int esz = dtype == SHAPY_DTYPE_BF16 ? 2 : 4;
  const float *wf32 = reinterpret_cast<const float *>(weights);   // biases + stem weights
  Lanes *L = nullptr;
  // The side streams and their fork / join events are one set per device, shared by every
  // caller: the whole issue of a forward (and a graph capture of it) holds this lock, so two host
  // threads or two caller streams cannot interleave their fork / join records on them.
  std::unique_lock<std::mutex> run_lock(g_run_mu, std::defer_lock);
  if (multi_stream) {
    run_lock.lock();
    int rc = get_lanes(&L);
    if (rc) return rc;
  }
  bool forked[N_SIDE] = {}, dirty[N_SIDE] = {};
  auto join_all = [&]() -> int {
    for (int i = 0; i < N_SIDE; ++i)
      if (dirty[i]) {
        SHAPY_HIP_TRY(hipEventRecord(L->join[i], L->s[i]));
        SHAPY_HIP_TRY(hipStreamWaitEvent(main, L->join[i], 0));
        dirty[i] = false;
      }
    return SHAPY_OK;
  };
  bool fork_recorded = false;
  for (int idx = 0; idx < n_ops; ++idx) {
    const ShapyOp &o = ops[idx];
    hipStream_t s = main;
    if (multi_stream) {
      if (o.barrier_before) {
        int rc = join_all();
        if (rc) return rc;
        for (int i = 0; i < N_SIDE; ++i) forked[i] = false;
        // The fork point of the new epoch is HERE, before any of its ops is enqueued on the main
        // stream.  (Round 2 recorded it lazily at the first side-lane op -- in plan order that
        // comes after the main lane's own ops of the epoch, so the side lanes waited for those
        // too: 100-190 us of a nearly idle chip per epoch, profiles/r03c_timeline_multistream.txt.)
        SHAPY_HIP_TRY(hipEventRecord(L->fork, main));
        fork_recorded = true;
      }
      if (o.lane > 0 && o.lane <= N_SIDE) {
        const int li = o.lane - 1;
        {
          const int rc = lane_stream(L, li, &s);
          if (rc) return rc;
        }
        if (!forked[li]) {
          if (!fork_recorded) {
            SHAPY_HIP_TRY(hipEventRecord(L->fork, main));
            fork_recorded = true;
          }
          SHAPY_HIP_TRY(hipStreamWaitEvent(s, L->fork, 0));
          forked[li] = true;
        }
        dirty[li] = true;
      }
    }
    // explicit dependencies on ops of OTHER lanes (the plan's happens-before graph: ops of one lane
    // are ordered by their stream): wait for the producers' events before this op ...
    // (a launch group starts all its members at once: the waits of EVERY member come first)
    if (multi_stream) {
      const int n_members = (o.type == SHAPY_OP_CONV && o.group > 1) ? o.group : 1;
      if (idx + n_members > n_ops) return SHAPY_EINVAL;
      for (int m = 0; m < n_members; ++m)
        for (int w = 0; w < 3; ++w) {
          const int e = ops[idx + m].wait[w];
          if (e >= 0) {
            if (e >= N_EVENTS) return SHAPY_EINVAL;
            SHAPY_HIP_TRY(hipStreamWaitEvent(s, L->ev[e], 0));
          }
        }
    }
    const int idx_first = idx;
    auto buf = [&](int64_t off) -> char * {
      return off < 0 ? nullptr : (char *)ws + off * (int64_t)B * esz;
    };
    auto make_desc = [&](const ShapyOp &q, ShapyConv &d) {
      d.in = buf(q.in_off);
      d.wgt = (const char *)weights + q.wgt_off * esz;
      d.bias = q.bias_off >= 0 ? wf32 + q.bias_off : nullptr;
      // a float32 plan may hold single layers in the bf16x6 arithmetic (their weights are split planes)
      d.dtype = (dtype == SHAPY_DTYPE_F32 && (q.tile & SHAPY_TILE_X6)) ? SHAPY_DTYPE_F32X6 : dtype;
      d.res = buf(q.res_off);
      d.out = buf(q.out_off);
      d.B = B; d.Hi = q.Hi; d.Wi = q.Wi; d.Cin = q.Cin; d.in_ld = q.in_ld;
      d.Ho = q.Ho; d.Wo = q.Wo; d.Cout = q.Cout; d.ksize = q.ksize; d.stride = q.stride;
      d.pad = q.pad; d.out_ld = q.out_ld; d.out_coff = q.out_coff; d.res_ld = q.res_ld;
      d.res_coff = q.res_coff; d.relu = q.relu; d.ups = q.ups; d.tile = q.tile;
      d.split_kib = 0;
      d.split_cnt_n = 0;
      d.wgt_wino = (d.dtype == SHAPY_DTYPE_F32 && q.wino_off >= 0) ? wf32 + q.wino_off : nullptr;
      // split-K layers: slab in the workspace, arrival counters in the caller's counter array
      d.split_ws = buf(q.split_off);
      if (d.split_ws && q.split_floats > 0 && q.split_off + q.split_floats <= ws_per_img) {
        const int64_t kib = q.split_floats * (int64_t)B * esz / 1024;     // (elements of the workspace's type)
        d.split_kib = kib > 0x7fffffff ? 0x7fffffff : (int32_t)kib;
      }
      if (counters && q.cnt_off >= 0 && q.cnt_n > 0 && q.cnt_off + q.cnt_n <= cnt_per_img) {
        d.split_cnt = counters + q.cnt_off * (int64_t)B;
        const int64_t n = q.cnt_n * (int64_t)B;
        d.split_cnt_n = n > 0x7fffffff ? 0x7fffffff : (int32_t)n;
      } else {
        d.split_cnt = nullptr;
      }
    };
    if (o.type == SHAPY_OP_CONV) {
      int rc = SHAPY_OK;
      if (o.group > 1) {
        // `group` independent layers of one depth level (same lane): ONE persistent F(4x4) launch
        // when all of them qualify, else one launch each on this stream
        const int n = o.group;
        if (n > 4 || idx + n > n_ops) return SHAPY_EINVAL;
        ShapyConv ds[4];
        for (int k = 0; k < n; ++k) {
          if (ops[idx + k].type != SHAPY_OP_CONV || ops[idx + k].lane != o.lane ||
              (k > 0 && ops[idx + k].barrier_before))
            return SHAPY_EINVAL;
          make_desc(ops[idx + k], ds[k]);
        }
        rc = conv2d_group(ds, n, s);
        if (rc == SHAPY_EINVAL) {
          rc = SHAPY_OK;
          for (int k = 0; k < n && rc == SHAPY_OK; ++k) rc = conv2d(ds[k], s);
        }
        idx += n - 1;
      } else {
        ShapyConv d;
        make_desc(o, d);
        rc = conv2d(d, s);
      }
      if (rc) {
        if (multi_stream) join_all();        // leave no forked lane unjoined behind an error
        return rc;
      }
    } else if (o.type == SHAPY_OP_STEM) {
      // (Wo % 16: the kernel's 16-pixel segments must not straddle rows; W % 32 == 0 guarantees it)
      if (o.Cin != 3 || o.Cout != 64 || o.ksize != 3 || o.stride != 2 || o.Wo % 16) return SHAPY_EINVAL;
      const long npix = (long)B * o.Ho * o.Wo;
      // the stem's own weights stay float32 (wgt_off counts float32 elements for this op)
      const dim3 grid((unsigned)((npix + 127) / 128));   // 8 segments of 16 pixels per workgroup
      if (esz == 4)
        hipLaunchKernelGGL(stem_conv_kernel<float>, grid, dim3(256), 0, s, input, wf32 + o.wgt_off,
                           wf32 + o.bias_off, (float *)buf(o.out_off), B, H, W, o.Ho, o.Wo, o.out_ld);
      else
        hipLaunchKernelGGL(stem_conv_kernel<unsigned short>, grid, dim3(256), 0, s, input,
                           wf32 + o.wgt_off, wf32 + o.bias_off, (unsigned short *)buf(o.out_off), B,
                           H, W, o.Ho, o.Wo, o.out_ld);
      SHAPY_HIP_TRY(hipGetLastError());
    } else if (o.type == SHAPY_OP_MEANPOOL) {
      const long total = (long)B * o.Cin;
      const dim3 grid((unsigned)((total + 255) / 256));
      if (esz == 4)
        hipLaunchKernelGGL(mean_pool_kernel<float>, grid, dim3(256), 0, s,
                           (const float *)buf(o.in_off), features_out, o.Hi * o.Wi, o.Cin, o.in_ld,
                           total);
      else
        hipLaunchKernelGGL(mean_pool_kernel<unsigned short>, grid, dim3(256), 0, s,
                           (const unsigned short *)buf(o.in_off), features_out, o.Hi * o.Wi, o.Cin,
                           o.in_ld, total);
      SHAPY_HIP_TRY(hipGetLastError());
    } else {
      return SHAPY_EINVAL;
    }
    // ... and signal this op's own event when a later op of another lane waits for it
    if (multi_stream)
      for (int q = idx_first; q <= idx; ++q)
        if (ops[q].sig >= 0) {
          if (ops[q].sig >= N_EVENTS) return SHAPY_EINVAL;
          SHAPY_HIP_TRY(hipEventRecord(L->ev[ops[q].sig], s));
        }
  }
  if (multi_stream) {
    const int rc = join_all();
    if (rc) return rc;
  }
  return SHAPY_OK;
}

}  // namespace shapy
