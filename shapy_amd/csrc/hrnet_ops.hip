// HRNet op-list executor + the two non-GEMM kernels of the backbone (stem conv, spatial mean).
// Reference: regressor/human_shape/models/backbone/hrnet.py:426-498.
#include <stdlib.h>

#include <mutex>

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
    for (int i = 0; i < 8; ++i) r[i] = fmaxf(acc[px][i >> 1][i & 1] + bs[i], 0.f);
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

// ---- upsample terms of a fuse output in ONE pass (hrnet.py:181-191) ----
// out = [relu](base [+ extra] [+ up2(y1) [+ up4(y2) [+ up8(y3)]]]), nearest upsampling, terms added in
// this order.  base / out: [B, H, W] rows of `ld` elements (they may be the same tensor: a thread reads
// and writes only its own 16 bytes); extra: dense [B, H, W, C] (the separately accumulated stride-2
// terms of the output, fuse_add = 2 plans); y_t: dense [B, H >> (t+1), W >> (t+1), C].  A thread owns 16 bytes of a pixel
// (4 floats / 8 bf16): every access is a full-width vector access, consecutive lanes are contiguous.
// Replaces the upsample-scatter epilogue of the conv kernel for these layers (fuse_add plans): there the
// FEW workgroups of the low-resolution GEMM (49 for the 7 x 7 source of a stage-4 module at B = 64, 196
// for 14 x 14) each re-read and re-write UPS x UPS times their tile of the output, and the whole output
// travels once per term -- 38 + 67 + 86 us for the three terms of the 56 x 56 output of a stage-4
// module at B = 64 (profiles/r04o_timeline_*), a 20 us job by traffic.
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
  static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
    *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <>
struct Vec16<unsigned short> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const unsigned short *p, float (&v)[8]) {
    const u32x4 x = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(x[e] << 16);
      v[2 * e + 1] = __uint_as_float(x[e] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(unsigned short *p, const float (&v)[8]) {
    store_row8(p, v);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void fuse_add_kernel(const T *base, T *out,   // (may be one tensor)
                                                       const T *__restrict__ extra,
                                                       const T *__restrict__ y1,
                                                       const T *__restrict__ y2,
                                                       const T *__restrict__ y3, long n_vec, int H,
                                                       int W, int C, int base_ld, int base_coff,
                                                       int out_ld, int out_coff, int relu) {
  constexpr int N = Vec16<T>::N;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_vec) return;
  const int cv = C / N;                       // 16-byte vectors per pixel
  const int c = (int)(i % cv) * N;
  const long pix = i / cv;
  const int x = (int)(pix % W);
  const long q = pix / W;
  const int y = (int)(q % H);
  const long b = q / H;
  float acc[N], t[N];
  Vec16<T>::load(base + pix * base_ld + base_coff + c, acc);
  if (extra) {
    Vec16<T>::load(extra + pix * (long)C + c, t);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] += t[e];
  }
  const T *ys[3] = {y1, y2, y3};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!ys[k]) break;
    const int sh = k + 1;
    const int Hk = H >> sh, Wk = W >> sh;
    Vec16<T>::load(ys[k] + ((b * Hk + (y >> sh)) * Wk + (x >> sh)) * (long)C + c, t);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] += t[e];
  }
  if (relu) {
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = fmaxf(acc[e], 0.f);
  }
  Vec16<T>::store(out + pix * out_ld + out_coff + c, acc);
}

// ---- side streams for the independent branches of a HighResolutionModule ----
constexpr int N_SIDE = 6;        // side streams: lanes 1..3 = branches, 4..6 = auxiliary chains
constexpr int N_EVENTS = 64;     // dependency events (ShapyOp.sig / .wait), reused from epoch to epoch
struct Lanes {
  hipStream_t s[N_SIDE] = {};
  hipStream_t s0 = nullptr;        // SHAPY_LANE_CU_EIGHTHS experiment: lane 0 on a CU-masked stream of its own
  hipEvent_t s0_in = nullptr, s0_out = nullptr;
  hipEvent_t fork = nullptr;
  hipEvent_t join[N_SIDE] = {};
  hipEvent_t ev[N_EVENTS] = {};
  bool ready = false;
};
static Lanes g_lanes[16];
static std::mutex g_lanes_mu;

// Side stream of lane li + 1, created on first use: HIP spreads streams over a few hardware queues,
// so streams that no plan uses (the auxiliary lanes 4..6 of the dag_aux experiment) should not exist.
// (SHAPY_LANE_PRIO=0 turns the priorities off; +1.1..1.6 % end to end, run T of round 3) lane i + 1
// (the branch with the smaller maps = the longer chain of latency-bound launches) gets a higher
// stream priority than lane i.
// EXPERIMENT (off unless SHAPY_LANE_CU_EIGHTHS is set; prepared at the end of round 4, never run):
// "a,b,c,d" = how many eighths of the CUs the streams of lanes 0..3 may use (residue classes of the CU
// index mod 8, handed out in this order; 0 = no mask).  In the traces the first launch of a small-map
// lane waits 160-300 us for workgroup slots that the two large lanes' launches keep refilling
// (profiles/r04o_module_tails.txt); a partition gives every lane slots of its own.  With a share for
// lane 0 the executor runs lane 0 on an internal masked stream between a fork from and a join to the
// caller's stream.  Masked streams carry no stream priority.
static int lane_cu_share(int lane, uint32_t (&mask)[8]) {
  static const char *env = getenv("SHAPY_LANE_CU_EIGHTHS");
  if (!env || lane > 3) return 0;
  int share[4] = {0, 0, 0, 0};
  if (sscanf(env, "%d,%d,%d,%d", &share[0], &share[1], &share[2], &share[3]) < 1) return 0;
  int first = 0;
  for (int l = 0; l < lane; ++l) first += share[l] > 0 ? share[l] : 0;
  const int n = share[lane];
  if (n <= 0 || first + n > 8) return 0;
  for (int w = 0; w < 8; ++w) {
    mask[w] = 0;
    for (int b = 0; b < 32; ++b) {
      const int r = (w * 32 + b) & 7;
      if (r >= first && r < first + n) mask[w] |= 1u << b;
    }
  }
  return n;
}

static int lane_stream(Lanes *L, int li, hipStream_t *out) {
  if (!L->s[li]) {
    static const int prio_mode = getenv("SHAPY_LANE_PRIO") ? atoi(getenv("SHAPY_LANE_PRIO")) : 1;
    uint32_t mask[8];
    if (lane_cu_share(li + 1, mask)) {
      SHAPY_HIP_TRY(hipExtStreamCreateWithCUMask(&L->s[li], 8, mask));
    } else if (prio_mode) {
      int plo = 0, phi = 0;
      SHAPY_HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));    // phi = highest (numerically lowest)
      int pr = plo - (li % 3 + 1);
      if (pr < phi) pr = phi;
      SHAPY_HIP_TRY(hipStreamCreateWithPriority(&L->s[li], hipStreamNonBlocking, pr));
    } else {
      SHAPY_HIP_TRY(hipStreamCreateWithFlags(&L->s[li], hipStreamNonBlocking));
    }
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

int hrnet_run(const ShapyOp *ops, int n_ops, const void *weights, const float *input, void *ws,
              int64_t ws_per_img, float *features_out, int B, int H, int W, int multi_stream,
              int dtype, hipStream_t main) {
  const int esz = dtype == SHAPY_DTYPE_BF16 ? 2 : 4;
  const float *wf32 = reinterpret_cast<const float *>(weights);   // biases + stem weights
  Lanes *L = nullptr;
  // The side streams and their fork / join events are one set per device, shared by every
  // caller: the whole issue of a forward (and a graph capture of it) holds this lock, so two host
  // threads or two caller streams cannot interleave their fork / join records on them.
  static std::mutex run_mu;
  std::unique_lock<std::mutex> run_lock(run_mu, std::defer_lock);
  if (multi_stream) {
    run_lock.lock();
    int rc = get_lanes(&L);
    if (rc) return rc;
  }
  // lane 0 = the caller's stream -- unless the CU-partition experiment gives lane 0 a share: then an
  // internal masked stream that starts behind the caller's stream here and is joined to it at the end
  hipStream_t caller = main;
  if (multi_stream) {
    uint32_t mask[8];
    if (lane_cu_share(0, mask)) {
      if (!L->s0) {
        SHAPY_HIP_TRY(hipExtStreamCreateWithCUMask(&L->s0, 8, mask));
        SHAPY_HIP_TRY(hipEventCreateWithFlags(&L->s0_in, hipEventDisableTiming));
        SHAPY_HIP_TRY(hipEventCreateWithFlags(&L->s0_out, hipEventDisableTiming));
      }
      SHAPY_HIP_TRY(hipEventRecord(L->s0_in, caller));
      SHAPY_HIP_TRY(hipStreamWaitEvent(L->s0, L->s0_in, 0));
      main = L->s0;
    }
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
  auto leave = [&]() -> int {            // internal lane-0 stream -> the caller's stream
    if (main != caller) {
      SHAPY_HIP_TRY(hipEventRecord(L->s0_out, main));
      SHAPY_HIP_TRY(hipStreamWaitEvent(caller, L->s0_out, 0));
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
      d.dtype = dtype;
      d.res = buf(q.res_off);
      d.out = buf(q.out_off);
      d.B = B; d.Hi = q.Hi; d.Wi = q.Wi; d.Cin = q.Cin; d.in_ld = q.in_ld;
      d.Ho = q.Ho; d.Wo = q.Wo; d.Cout = q.Cout; d.ksize = q.ksize; d.stride = q.stride;
      d.pad = q.pad; d.out_ld = q.out_ld; d.out_coff = q.out_coff; d.res_ld = q.res_ld;
      d.res_coff = q.res_coff; d.relu = q.relu; d.ups = q.ups; d.tile = q.tile;
      d.reserved0 = 0;
      d.wgt_wino = (dtype == SHAPY_DTYPE_F32 && q.wino_off >= 0) ? wf32 + q.wino_off : nullptr;
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
        if (multi_stream) {                  // leave no forked lane unjoined behind an error
          join_all();
          leave();
        }
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
    } else if (o.type == SHAPY_OP_FUSEADD) {
      // Ho x Wo x Cout output; res = the base tensor; wino_off = a second full-resolution term (-1:
      // none); in_off / wgt_off / bias_off = the up to three low-resolution terms -- all offsets into
      // the WORKSPACE (upsample factors 2, 4, 8; ksize = how many, 0..3)
      const int nv = esz == 4 ? 4 : 8;
      if (o.ksize < 0 || o.ksize > 3 || o.res_off < 0 || o.out_off < 0 || (o.ksize > 0 && o.in_off < 0) ||
          (o.ksize > 1 && o.wgt_off < 0) || (o.ksize > 2 && o.bias_off < 0) || o.Cout % nv ||
          o.out_ld % nv || o.out_coff % nv || o.res_ld % nv || o.res_coff % nv ||
          (o.Ho & ((1 << o.ksize) - 1)) || (o.Wo & ((1 << o.ksize) - 1)))
        return SHAPY_EINVAL;
      const long n_vec = (long)B * o.Ho * o.Wo * (o.Cout / nv);
      const dim3 grid((unsigned)((n_vec + 255) / 256));
      char *y1 = o.ksize > 0 ? buf(o.in_off) : nullptr, *y2 = o.ksize > 1 ? buf(o.wgt_off) : nullptr,
           *y3 = o.ksize > 2 ? buf(o.bias_off) : nullptr, *ex = buf(o.wino_off);
      if (esz == 4)
        hipLaunchKernelGGL(fuse_add_kernel<float>, grid, dim3(256), 0, s, (const float *)buf(o.res_off),
                           (float *)buf(o.out_off), (const float *)ex, (const float *)y1, (const float *)y2,
                           (const float *)y3, n_vec, o.Ho, o.Wo, o.Cout, o.res_ld, o.res_coff, o.out_ld,
                           o.out_coff, o.relu);
      else
        hipLaunchKernelGGL(fuse_add_kernel<unsigned short>, grid, dim3(256), 0, s,
                           (const unsigned short *)buf(o.res_off), (unsigned short *)buf(o.out_off),
                           (const unsigned short *)ex, (const unsigned short *)y1, (const unsigned short *)y2,
                           (const unsigned short *)y3, n_vec, o.Ho, o.Wo, o.Cout, o.res_ld, o.res_coff,
                           o.out_ld, o.out_coff, o.relu);
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
    int rc = join_all();
    if (rc) return rc;
    rc = leave();
    if (rc) return rc;
  }
  return SHAPY_OK;
}

}  // namespace shapy
