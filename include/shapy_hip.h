/* shapy_hip.h -- C-ABI of the MI355X-native SHAPY hot path (libshapy_hip.so).
 *
 * Plain pointers and sizes only: no torch / ATen types cross this boundary.  Every pointer
 * is DEVICE memory unless its name ends in _host.  Every entry point enqueues its work on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream), never synchronises
 * the device and returns 0 on success or a hipError_t / negative SHAPY_E* code.
 *
 * What each entry point replaces in the reference (muelea/shapy):
 *
 *  shapy_conv2d / shapy_hrnet_run
 *      the 331 cuDNN convolutions + eval-mode BatchNorm + ReLU + residual adds + nearest
 *      upsampling + concat + spatial mean of HighResolutionNet.forward
 *      (regressor/human_shape/models/backbone/hrnet.py:426-498, :175-193)
 *  shapy_regressor_affine_f32
 *      IterativeRegression.forward / MLP.forward
 *      (regressor/human_shape/models/common/networks.py:536-592, :392-400)
 *  shapy_smplx_pose_f32, shapy_smplx_skin_f32, shapy_smplx_joints_f32, shapy_smplx_forward_f32
 *      ContinuousRotReprDecoder.forward (models/common/pose_utils.py:138-153),
 *      batch_rodrigues (utils/rotation_utils.py:5-37), lbs() and batch_rigid_transform
 *      (models/body_models/lbs.py:99-196, :242-295), the landmark code (lbs.py:20-94),
 *      WeakPerspectiveCamera.forward (models/camera/camera_projection.py:181-213)
 *  shapy_mesh_to_mesh_f32 (+ _workspace_bytes), shapy_mesh_to_mesh_f64
 *      mesh_mesh_intersect_cuda.mesh_to_mesh_forward
 *      (mesh-mesh-intersection/src/mesh_mesh_intersect.cpp:36-64,
 *       src/mesh_mesh_intersect_cuda_op.cu:969-1079 and every kernel it launches)
 *  shapy_body_measure_f32 (+ _workspace_bytes)
 *      BodyMeasurements.forward: compute_mass / compute_height / compute_peripheries incl.
 *      the per-mesh scipy ConvexHull (mesh-mesh-intersection/body_measurements/
 *      body_measurements.py:99-246)
 *  shapy_hrnet_graph_create / _launch / _destroy
 *      the same forward as shapy_hrnet_run, captured once into a hipGraph (no reference
 *      counterpart: the reference launches every cuDNN call from Python)
 *  shapy_pose_decode_f32, shapy_weak_persp_project_f32, shapy_joint_regress_f32
 *      the decoders / camera / extra-joint regressors used stand-alone by the host mirror
 *      (pose_utils.py:84-153, camera_projection.py:181-213, body_models.py:738-744)
 *  shapy_b2a_polynomial_f32
 *      B2A / Polynomial.forward (attributes/attributes/attributes_betas/polynomial.py:61-69,137-140)
 *  shapy_crop_resize_normalize_u8
 *      the CPU crop + resize + normalise of the OpenPose dataset
 *      (regressor/human_shape/data/datasets/openpose.py:146-246, utils/transf_utils.py:53-96)
 *  shapy_aligned_point_error_f32, shapy_p2p_error_f64
 *      PointError / the alignments / v2vhdError of the evaluator
 *      (regressor/human_shape/utils/metrics.py:31-56,84-277,335-460)
 */
#ifndef SHAPY_HIP_H
#define SHAPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHAPY_OK 0
#define SHAPY_EINVAL (-1)      /* bad argument (shape / alignment / unsupported mode) */
#define SHAPY_EWORKSPACE (-2)  /* workspace too small */

/* ABI version; bumped whenever a struct below changes. */
int shapy_abi_version(void);
/* name of the gfx target the library was built for ("gfx950") */
const char *shapy_build_arch(void);

/* ---------------------------------------------------------------------------------------
 * Convolution / GEMM (implicit GEMM on f32 MFMA, NHWC activations, OHWI weights)
 *
 *   out[b,ho,wo, out_coff + n] = act( bias[n] + sum_{kh,kw,c} in[b, ho*s-p+kh, wo*s-p+kw, c]
 *                                                   * wgt[n, kh, kw, c]  (+ res[...]) )
 * With ups > 1 the value computed for the (low-res) output pixel is written to the ups x ups
 * block of pixels of an output (and residual) tensor of size [B, Ho*ups, Wo*ups] -- the
 * conv1x1 + BN + nearest-Upsample + add of an HRNet fuse layer (hrnet.py:125-136,184-191) in
 * one pass.  BatchNorm is folded into wgt/bias by the host (float64).
 * A plain GEMM out[M,N] = in[M,K] * wgt[N,K]^T + bias is the case ksize=1, Hi=Wi=Ho=Wo=1,
 * B=M.  Requirements: Cin % 16 == 0 (f32) / % 32 == 0 (bf16), in and wgt 16-byte aligned,
 * in_ld a multiple of 16 bytes; tensors smaller than 2 GiB (32-bit buffer offsets).
 * ------------------------------------------------------------------------------------- */
/* F32X6: float32 storage everywhere (same tensors as F32), products formed on the bf16 matrix
 * cores from the exact 3-way bf16 split of both operands (6 MFMAs per product, f32
 * accumulation; csrc/conv_x6.hip) -- float32-class accuracy, Cin only needs to be a multiple
 * of 4.  In this mode ShapyConv.wgt holds the weights already split: bfloat16
 * [Cout][3][Kp], planes (h, m, l) with h + m + l == w exactly, K = ksize*ksize*Cin zero-padded
 * to Kp = a multiple of 32 (shapy_amd/utils/split.py:split_bf16x3). */
enum { SHAPY_DTYPE_F32 = 0, SHAPY_DTYPE_BF16 = 1, SHAPY_DTYPE_F32X6 = 2 };

typedef struct ShapyConv {
  const void *in;     /* [B, Hi, Wi, in_ld]   (first Cin channels of each pixel are used)   */
  const void *wgt;    /* [Cout, ksize, ksize, Cin]                                         */
  const float *bias;  /* [Cout] float32 or NULL                                            */
  const void *res;    /* residual, indexed like out (res_ld, res_coff), or NULL; may == out */
  void *out;          /* [B, Ho*ups, Wo*ups, out_ld]                                       */
  int32_t B, Hi, Wi, Cin, in_ld;
  int32_t Ho, Wo, Cout;
  int32_t ksize, stride, pad;
  int32_t out_ld, out_coff, res_ld, res_coff;
  int32_t relu;       /* 1: ReLU after the (residual) add                                  */
  int32_t ups;        /* 1 = none; 2/4/8 = nearest-upsample scatter                        */
  int32_t tile;       /* 0 = choose automatically; low byte: a SHAPY_TILE_* id; higher bits are
                         A/B knobs of tools/conv_bench.py (shapy_amd/_lib.py: TILES), e.g. 0x2000
                         never Winograd, 0x4000 / 0x8000 Winograd tile groups, 0x20000 Winograd K
                         loop chunk by chunk, 0x40000 / 0x80000 three / one chunk(s) of loads in
                         flight.  Speed only: every setting computes the same convolution.
                         Two fields describe DATA / arithmetic instead: SHAPY_TILE_WINO4 (0x100000)
                         says that wgt_wino holds F(4x4,3x3) filters (below); SHAPY_TILE_KSPLIT(S)
                         (bits 21..22 = S - 1, S = 1..4) runs the layer with its K loop cut into S
                         slices on S workgroups per output tile (split_ws / split_cnt below; the S
                         partial sums are added in slice order, so the result is deterministic but
                         differs from S = 1 in the last bits).  Available on the F(4x4) kernel (Cin / 16
                         divisible by S) and on the implicit-GEMM kernel (ups == 1, Cin % 32 == 0 in bf16,
                         tiles up to 64 x 64; every slice must get at least one K chunk); anything
                         else is SHAPY_EINVAL.                                                   */
  int32_t dtype;      /* storage type of in / wgt / res / out: SHAPY_DTYPE_F32 (f32 MFMA, exact
                         f32) or SHAPY_DTYPE_BF16 (bf16 MFMA, f32 accumulate; Cin % 8 == 0, and
                         Cin >= 32 with ups == 1 when Cin % 32 != 0: the flat-K kernel)          */
  int32_t split_kib;  /* SHAPY_TILE_KSPLIT(S > 1): capacity of split_ws in KiB (the launch is refused, SHAPY_EINVAL,
                         when it would need more); 0 otherwise                                  */
  const void *wgt_wino; /* NULL, or the Winograd F(2x2,3x3) transform of wgt for a float32
                         3x3 / stride 1 / pad 1 layer: U[p = 4i+j][Cin/16][Cout][16] float32,
                         U[i][j] = (G g G^T)[i][j] (shapy_amd/utils/winograd.py).  When given
                         (and Cin % 16 == 0, Cout % 48 == 0 or % 64 == 0, 16-byte aligned out / res / bias
                         rows) the layer runs on csrc/conv_wino.hip: 2.25x fewer MFMAs, result
                         equal to the direct sum up to float32 rounding of the transforms.
                         With SHAPY_TILE_WINO4 set in `tile`: the F(4x4,3x3) transform instead,
                         U[p = 6i+j][Cin/16][Cout][16] with the 6x3 G of the points {0, +-1, +-2,
                         inf} (winograd.transform_filters4); needs Cout % 48 == 0; runs on
                         csrc/conv_wino4.hip (4x fewer MFMAs than the direct sum).  Tensors
                         beyond 1 GiB (the kernel's 32-bit offset scheme) run the direct kernel
                         on `wgt` instead; SHAPY_EINVAL when the layer shape does not qualify.  */
  void *split_ws;     /* SHAPY_TILE_KSPLIT(S > 1) only (ABI 8): scratch for the partial sums, 16-byte
                         aligned, private to this call while it runs.  F(4x4): S * B * ceil(Hi/4) *
                         ceil(Wi/4) * 16 * Cout floats; implicit GEMM: S * Mpad * Npad floats with M = B * Ho
                         * Wo and Cout rounded up to the tile the library picks (at most 64 x 64 for such
                         layers: Mpad <= M + 63, Npad <= Cout + 63)                                */
  int32_t *split_cnt; /* ... and int32 arrival counters that are ZERO when the call starts; the kernel
                         leaves them zero (so the same counters serve the next call on the same stream,
                         but must be zeroed again after a failed / aborted launch).  F(4x4): 2 * ceil(B *
                         ceil(Hi/4) * ceil(Wi/4) / 16) * (Cout / 16); implicit GEMM: 8 per tile, at most
                         8 * ceil(M / 32) * ceil(Cout / 48)                                         */
  int32_t split_cnt_n; /* capacity of split_cnt in int32 (checked like split_kib)                   */
} ShapyConv;
#define SHAPY_TILE_WINO4 0x100000
#define SHAPY_TILE_KSPLIT(s) ((((s) - 1) & 3) << 21)
#define SHAPY_TILE_W4_KSPLIT(s) SHAPY_TILE_KSPLIT(s)
/* In a FLOAT32 op list (ShapyOp.tile): this layer's weights are the three bf16 planes [Cout][3][Kp] of the exact
 * 3-way split and its products come from the bf16 matrix cores (the arithmetic SHAPY_DTYPE_F32X6 selects for a
 * whole plan; tensors stay float32).  A direct shapy_conv2d call says the same with ShapyConv.dtype. */
#define SHAPY_TILE_X6 0x800000

int shapy_conv2d(const ShapyConv *desc_host, void *stream);

/* Up to four INDEPENDENT F(4x4,3x3) Winograd layers in one persistent launch
 * (csrc/conv_wino4g.hip): the convolutions at the same depth of the parallel branches of a
 * HighResolutionModule -- the reference walks them one branch after the other
 * (regressor/human_shape/models/backbone/hrnet.py:175-179, `for i in range(self.num_branches):
 * x[i] = self.branches[i](x[i])`).  Every descriptor must be one that shapy_conv2d would run on
 * the F(4x4) kernel (float32, SHAPY_TILE_WINO4 set, wgt_wino given, Cin % 16 == 0, Cout % 48 == 0,
 * tensors within 1 GiB); otherwise SHAPY_EINVAL and nothing is launched -- call shapy_conv2d per
 * layer instead.  No descriptor may read what another one writes.  Results equal those of n
 * shapy_conv2d calls. */
int shapy_conv2d_group(const ShapyConv *descs_host, int n, void *stream);

/* ---------------------------------------------------------------------------------------
 * HRNet op list.  The host (Python) flattens the module tree into `ops`; buffers are
 * offsets (in floats, per image) into one workspace allocation that is scaled by B.
 * ------------------------------------------------------------------------------------- */
enum { SHAPY_OP_CONV = 0, SHAPY_OP_STEM = 1, SHAPY_OP_MEANPOOL = 2 };
typedef struct ShapyOp {
  int32_t type;
  int32_t lane;                /* stream id (0..6; 0 = the caller's stream): ops of different
                                  lanes may run concurrently unless ordered by a barrier or by
                                  sig / wait                                                 */
  int32_t barrier_before;      /* 1: all lanes must have finished before this op starts     */
  int32_t Hi, Wi, Cin, in_ld, Ho, Wo, Cout, ksize, stride, pad;
  int32_t out_ld, out_coff, res_ld, res_coff, relu, ups, tile;
  int32_t group;               /* > 1 on the FIRST of `group` consecutive CONV ops that are independent
                                  of each other (same lane, same epoch): they are issued as one
                                  shapy_conv2d_group launch when every one is an F(4x4) layer,
                                  otherwise one after the other; 0 / 1 elsewhere             */
  int32_t sig;                 /* >= 0: event slot (0..63) recorded on this op's stream after it: a
                                  later op of ANOTHER lane waits for it; -1: none             */
  int32_t wait[3];             /* event slots to wait for before this op (-1 = unused): its
                                  producers on other lanes.  Ops of one lane are ordered by the
                                  lane's stream; lanes 1..6 are side streams, 0 the caller's.
                                  With sig / wait a plan needs barrier_before only where it
                                  wants every lane joined (slots are reused after a barrier)   */
  int64_t in_off, out_off, res_off;     /* per-image float offsets into the workspace; -1 = none;
                                            in_off == -2: the network input (STEM)           */
  int64_t wgt_off, bias_off;            /* float offsets into the weight blob; -1 = none     */
  int64_t wino_off;                     /* float offset of the Winograd-transformed filters
                                           (ShapyConv.wgt_wino) in the blob; -1 = none        */
  int64_t split_off;                    /* split-K layers (SHAPY_TILE_KSPLIT in `tile`, ABI 8):
                                           per-image float offset of ShapyConv.split_ws in the
                                           workspace; -1 = none                                */
  int64_t cnt_off;                      /* ... and per-image int32 offset of ShapyConv.split_cnt
                                           in the `counters` argument; -1 = none               */
  int64_t split_floats, cnt_n;          /* per-image capacities of the two, the slab's in workspace
                                           ELEMENTS (float32 or bf16) (the executor hands
                                           B times these to the kernel launcher as split_kib /
                                           split_cnt_n)                                         */
} ShapyOp;

/* input: [B,3,H,W] NCHW f32 (the reference's layout, iterative_regressor.py:623);
 * features_out: [B, Cfeat].  workspace must hold B * ws_floats_per_image floats.
 * counters (ABI 8): B * cnt_per_image int32 for the split-K layers of the plan (ShapyOp.cnt_off), or
 * NULL when cnt_per_image == 0.  The caller zeroes them ONCE after allocating them; every completed
 * forward leaves them zero.  Like the workspace they belong to one forward at a time. */
int shapy_hrnet_run(const ShapyOp *ops_host, int n_ops, const void *weights,
                    const float *input_nchw, void *workspace, int64_t ws_elems_per_image,
                    int32_t *counters, int64_t cnt_per_image, float *features_out, int B, int H,
                    int W, int multi_stream, int dtype, void *stream);

/* The executor's side stream of `lane` (1..6; lanes 1..3 carry the branches of a HighResolutionModule)
 * on the current device, created on first use and owned by the library.  For host-side work that should
 * ride on a stream the process already has -- the RCCL all-gather of the predicted betas in a
 * data-parallel step (shapy_amd/parallel.py, mode 'lane') -- instead of adding another one.  Work put
 * there is ordered with the lane's ops of later forwards like any stream work; the caller orders it
 * against its own stream with events. */
int shapy_hrnet_lane_stream(int lane, void **stream_out);

/* The same forward captured once into a hipGraph (side-stream branches included) and replayed
 * with one launch: removes the ~330 kernel launches + fork/join events per forward from the
 * host, which dominate at small batch.  All pointers are baked into the graph: the caller
 * keeps input / workspace / features_out / weights alive and at the same addresses for the
 * life of the graph, copies new images into `input_nchw` before shapy_hrnet_graph_launch and
 * reads `features_out` after it (stream-ordered).  ops_host is only read during create. */
int shapy_hrnet_graph_create(const ShapyOp *ops_host, int n_ops, const void *weights,
                             const float *input_nchw, void *workspace,
                             int64_t ws_elems_per_image, int32_t *counters, int64_t cnt_per_image,
                             float *features_out, int B, int H, int W, int multi_stream, int dtype,
                             void **graph_out);
int shapy_hrnet_graph_launch(void *graph, void *stream);
int shapy_hrnet_graph_destroy(void *graph);
/* dtype = SHAPY_DTYPE_F32: weights / workspace are float32 (the parity path);
 * SHAPY_DTYPE_BF16: conv weights and activations are bfloat16 (biases stay float32 and live in
 * the blob at 4-byte granularity: bias_off counts float32 elements, wgt_off bfloat16 elements),
 * the network input and the features stay float32 (BASELINE configs[2]). */

/* ---------------------------------------------------------------------------------------
 * Iterative regressor, affine-collapsed form.  The SHAPY_A MLP has no activation and no
 * normalisation (configs/b2a_expose_hrnet_demo.yaml:200-207), so each stage is
 *     p_i = p_{i-1} + Wf * feat + Wp * p_{i-1} + b
 * with Wf [P, F], Wp [P, P], b [P] collapsed from the three Linear layers by the host in
 * float64.  params_out: [num_stages, B, P].
 * ------------------------------------------------------------------------------------- */
int shapy_regressor_affine_f32(const float *features, const float *Wf, const float *Wp,
                               const float *bias, const float *mean_param, float *params_out,
                               int B, int F, int P, int num_stages, int cond_per_body,
                               void *stream);
/* mean_param: [P] (cond_per_body = 0) or a per-body initial condition [B,P] (= 1), the
 * `cond` argument of IterativeRegression.forward (networks.py:536-566). */

/* The same regressor with the stages collapsed as well (every body starts from the same mean,
 * i.e. no `cond`):  with A = I + Wp,  p_s = A^s mean + (sum_{k<s} A^k)(Wf feat + b), so
 * params_out[s] = W_all[s] feat + b_all[s] with W_all [num_stages*P, F], b_all [num_stages*P]
 * folded by the host in float64 (shapy_amd/models/common/networks.py).  One launch. */
int shapy_regressor_collapsed_f32(const float *features, const float *W_all, const float *b_all,
                                  float *params_out, int B, int F, int P, int num_stages,
                                  void *stream);

/* ---------------------------------------------------------------------------------------
 * SMPL-X
 * ------------------------------------------------------------------------------------- */
typedef struct ShapySmplxModel {
  int32_t V, J, NB;             /* vertices, joints (55), shape comps incl. expression      */
  int32_t P;                    /* pose-feature dim = (J-1)*9, Ppad = round_up(P,32)         */
  int32_t Ppad, NBpad;          /* K paddings of the two GEMMs                               */
  int32_t n_static_lmk, n_dyn_lmk, n_dyn_rows, n_neck;
  const int32_t *parents;       /* [J], parents[0] = -1                                      */
  const float *J_template;      /* [J,3]      = J_regressor * v_template      (float64 fold) */
  const float *J_shapedirs;     /* [J,3,NB]   = J_regressor * shapedirs                      */
  const float *v_template;      /* [V*3]                                                     */
  const float *shapedirs_t;     /* [V*3 (row-padded to x96), NBpad]  K-contiguous            */
  const float *posedirs_t;      /* [V*3 (row-padded to x96), Ppad]   K-contiguous            */
  const float *lbs_weights_t;   /* [J, V]                                                    */
  const int32_t *faces;         /* [F,3]                                                     */
  const int32_t *lmk_faces_idx;         /* [n_static_lmk]                                    */
  const float *lmk_bary;                /* [n_static_lmk,3]                                  */
  const int32_t *dyn_lmk_faces_idx;     /* [n_dyn_rows, n_dyn_lmk]                           */
  const float *dyn_lmk_bary;            /* [n_dyn_rows, n_dyn_lmk, 3]                        */
  const int32_t *neck_kin_chain;        /* [n_neck]                                          */
} ShapySmplxModel;

enum { SHAPY_POSE_ROTMAT = 0, SHAPY_POSE_CONT6D = 1, SHAPY_POSE_AXIS_ANGLE = 2 };

/* Per body: decode the pose (n_pose joints given, the remaining joints are identity), regress
 * the rest joints from the shape coefficients, run the kinematic chain.
 *   pose:        [B, n_pose, 6] (CONT6D, interleaved [a1x,a2x,a1y,a2y,a1z,a2z]) | [B,n_pose,3]
 *                (AXIS_ANGLE) | [B,n_pose,3,3] (ROTMAT)
 *   coeffs:      [B, NBpad]  shape (+expression) coefficients, zero padded
 *   rot_out:     [B, J, 3, 3]   pose_feat_out: [B, Ppad]   A_out: [B, J, 12] (3x4 rel. transforms)
 *   joints_out:  [B, J, 3] posed joints     dyn_row_out: [B] int32 LUT row (lbs.py:33-41)   */
int shapy_smplx_pose_f32(const ShapySmplxModel *model_host, const float *pose, int pose_type,
                         int n_pose, const float *coeffs, float *rot_out, float *pose_feat_out,
                         float *A_out, float *joints_out, int32_t *dyn_row_out, int B,
                         void *stream);

/* The glue between the regressor's parameter vectors and the SMPL-X kernels in one launch
 * (iterative_regressor.py:646-660, body_models.py:660-700): params [S,B,P] (all stages) ->
 *   rot_out    [S,B,n_joints,3,3]  decoded poses of every stage; the pose parameters are the
 *              n_joints * (6 | 3) floats starting at pose_off of every parameter vector
 *   coeffs_out [B,NBpad]           betas of the LAST stage (n_betas floats at betas_off), zero padded
 *   cam_out    [B,3] or NULL       camera parameters of the last stage (3 floats at cam_off) */
int shapy_head_prepare_f32(const float *params, int S, int B, int P, int pose_off, int n_joints,
                           int pose_type, int betas_off, int n_betas, int NBpad, int cam_off,
                           float *rot_out, float *coeffs_out, float *cam_out, void *stream);

/* Stand-alone decoders: [n,6] (CONT6D, pose_utils.py:138-153) or [n,3] (AXIS_ANGLE,
 * rotation_utils.py:5-37) -> [n,3,3]. */
int shapy_pose_decode_f32(const float *pose, int pose_type, float *rot_out, int64_t n,
                          void *stream);

/* WeakPerspectiveCamera.forward (camera_projection.py:181-213): points [B,N,3], scale [B],
 * translation [B,2] -> out [B,N,2]. */
int shapy_weak_persp_project_f32(const float *points, const float *scale,
                                 const float *translation, float *out, int B, int N,
                                 int scale_first, void *stream);

/* vertices[b,v,:] = (sum_j W[v,j] A[b,j]) * [v_posed[b,v,:]; 1]   (lbs.py:187-190) */
int shapy_smplx_skin_f32(const ShapySmplxModel *model_host, const float *A, const float *v_posed,
                         float *vertices_out, int B, void *stream);

/* Extra joint regressor (body_models.py:738-744): out[b,j,:] = sum_v regressor[j,v] *
 * vertices[b,v,:];  regressor [Jn,V], vertices [B,V,3], out [B,Jn,3]. */
int shapy_joint_regress_f32(const float *regressor, const float *vertices, float *out, int B,
                            int V, int Jn, void *stream);

/* joints_out[b] = cat(posed joints [J], static landmarks, dynamic landmarks) [B, n_out, 3];
 * proj_out = softplus(cam[:,0]) * (joints_xy + cam[:,1:3])  [B, n_out, 2] (NULL to skip). */
int shapy_smplx_joints_f32(const ShapySmplxModel *model_host, const float *posed_joints,
                           const float *vertices, const int32_t *dyn_row, const float *camera,
                           float *joints_out, float *proj_out, float *cam_scale_out, int B,
                           int use_face_contour, void *stream);

/* Argument glue of SMPLX.forward (body_models.py:660-700) in one launch: up to 7 pose parts of
 * rotation matrices (device pointers in the HOST array parts_host, part k = [B, n_joints_host[k], 3, 3]
 * with contiguous joints and part_bstride_host[k] floats between consecutive bodies -- 0 = one row
 * broadcast to every body, NULL array = dense; slices such as rot[:, 1:] need no copy; a NULL part is
 * identity) concatenated into pose_out [B, sum n, 3, 3]; betas (rows of nb floats, betas_bstride apart;
 * 0 = one row for all) (+ expression, rows of ne floats, NULL = none) into coeffs_out [B, NBpad] (zero
 * padded) and, when coeffs_shape_out is given, the same row with the expression part zeroed. */
int shapy_smplx_prepare_f32(const float *const *parts_host, const int32_t *n_joints_host,
                            const int64_t *part_bstride_host, int n_parts, const float *betas,
                            int64_t betas_bstride, int nb, const float *expression,
                            int64_t expr_bstride, int ne, int NBpad, float *pose_out, float *coeffs_out,
                            float *coeffs_shape_out, int B, void *stream);

/* The whole SMPL-X layer (SMPLX.forward, models/body_models/body_models.py:628-767, on prepared
 * inputs) in ONE call: shape blend GEMM(s), pose decode + joint regression + kinematic chain, pose
 * blend GEMM, skinning, landmarks (+ weak-perspective projection when `camera` is given) -- the
 * launches of the entry points above, enqueued back to back.  coeffs_shape: the coefficients with the
 * expression part zeroed (then v_shaped receives the shape-only vertices), or NULL (v_shaped_full IS
 * v_shaped; v_shaped may be NULL).  shape_only: stop after the shape GEMM(s) (SMPL.forward_shape).
 * Buffers: v_shaped_full, v_shaped, v_posed, vertices [B,V,3]; rot [B,J,3,3]; pose_feat [B,Ppad];
 * A [B,J,12]; posed_joints [B,J,3]; dyn_row [B] int32; joints_out [B,n_out,3]; proj_out [B,n_out,2]
 * and cam_scale_out [B,1] or NULL. */
int shapy_smplx_forward_f32(const ShapySmplxModel *model_host, const float *pose, int pose_type,
                            int n_pose, const float *coeffs, const float *coeffs_shape,
                            const float *camera, float *v_shaped_full, float *v_shaped, float *rot,
                            float *pose_feat, float *A, float *posed_joints, int32_t *dyn_row,
                            float *v_posed, float *vertices, float *joints_out, float *proj_out,
                            float *cam_scale_out, int B, int use_face_contour, int shape_only,
                            void *stream);

/* ---------------------------------------------------------------------------------------
 * Mesh-mesh intersection (the reference's operator boundary)
 *   query  [B,Q,3,3] f32, target [B,F,3,3] f32
 *   faces_out int64 [B, Q*max_coll] (-1 = empty), bcs_out f32 [B, Q*max_coll, 2, 3] (0 = empty)
 * Slot order inside a query triangle's max_coll slots is ascending target-face index
 * (the reference's is BVH traversal order, unspecified).  Hits beyond max_coll are dropped
 * and counted in *overflow_out (int32 device counter, may be NULL); the reference writes
 * out of bounds in that case (.cu:551,565).
 * shapy_mesh_to_mesh_f64 is the reference's second instantiation (AT_DISPATCH_FLOATING_TYPES, .cu:996) on
 * float64 triangles: the same observable semantics, double arithmetic with the float constants the reference
 * keeps in it (CMP converts to float and compares against FLT_EPSILON, .cu:91-92; the 1e-4 determinant cut);
 * bcs_out is float64.  No SHAPY caller passes float64, so it is the brute-force scan for every Q (no
 * workspace): slow for large query meshes, correct for all; bit-exact against the oracle's float64 build.
 * ------------------------------------------------------------------------------------- */
size_t shapy_mesh_to_mesh_workspace_bytes(int B, int Q, int F, int max_coll);
int shapy_mesh_to_mesh_f32(const float *query, const float *target, int B, int Q, int F,
                           int max_coll, int64_t *faces_out, float *bcs_out, void *workspace,
                           size_t workspace_bytes, int32_t *overflow_out, void *stream);
int shapy_mesh_to_mesh_f64(const double *query, const double *target, int B, int Q, int F,
                           int max_coll, int64_t *faces_out, double *bcs_out, int32_t *overflow_out,
                           void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused virtual measurements: mass, height, chest, waist, hips from v_shaped + faces.
 *   lm_face[5], lm_bary[5][3]: HeadTop, HeelLeft, chest, waist, hips landmarks (host memory)
 *   out: [B,5] f32 = mass(kg), height, chest, waist, hips (m)
 * ------------------------------------------------------------------------------------- */
size_t shapy_body_measure_workspace_bytes(int B, int F, int max_coll);
int shapy_body_measure_f32(const float *v_shaped, const int32_t *faces, int B, int V, int F,
                           const int32_t *lm_face_host, const float *lm_bary_host, int max_coll,
                           float *out, void *workspace, size_t workspace_bytes,
                           int32_t *overflow_out, void *stream);

/* B2A attribute head: degree-2 polynomial of the betas followed by a Linear layer
 * (attributes/attributes/attributes_betas/polynomial.py:61-69,137-140).
 * betas [B,NB], weight [NA, NB + NB(NB+1)/2], bias [NA] -> out [B,NA]. */
int shapy_b2a_polynomial_f32(const float *betas, const float *weight, const float *bias,
                             float *out, int B, int NB, int NA, void *stream);

/* ---------------------------------------------------------------------------------------
 * Input pre-processing (the step in front of the hot path): crop window -> bilinear resize
 * to S x S -> clamp -> normalise, for a ragged batch of full HWC uint8 images.
 *   images: concatenated uint8 RGB images; img_off[b]: byte offset of image b; img_hw[b] = (H, W)
 *   boxes[b] = (ul_x, ul_y, br_x, br_y): integer crop window in full-image pixels, computed like
 *   transf_utils.transform (regressor/human_shape/utils/transf_utils.py:40-66); pixels of the
 *   window outside the image are zero.   out: [B,3,S,S] float32 = (clamp(x,0,1) - mean) / std.
 * ------------------------------------------------------------------------------------- */
int shapy_crop_resize_normalize_u8(const unsigned char *images, const int64_t *img_off,
                                   const int32_t *img_hw, const int32_t *boxes, float *out, int B,
                                   int S, const float *mean_host, const float *std_host,
                                   void *stream);

/* ---------------------------------------------------------------------------------------
 * Evaluator metrics (the step behind the hot path).
 * Aligned point error: PointError(alignment)(est, gt) of regressor/human_shape/utils/metrics.py
 * :335-365 with the alignments of :59-277 -- alignment 0 none, 1 translation, 2 scale,
 * 3 procrustes -- as used by Evaluator._compute_v2v / _compute_mpjpe
 * (regressor/human_shape/evaluation.py:120-225).   est, gt: [B,P,3] float32.
 * Any of err_out [B,P], err_mean_out [B], aligned_out [B,P,3] (the aligned estimate the
 * alignment objects return) may be NULL.
 * ------------------------------------------------------------------------------------- */
int shapy_aligned_point_error_f32(const float *est, const float *gt, int B, int P, int alignment,
                                  float *err_out, float *err_mean_out, float *aligned_out,
                                  void *stream);

/* Point-to-point error between meshes of different topology (P2P-20k): v2vhdError.__call__
 * (regressor/human_shape/utils/metrics.py:367-460; caller evaluation.py:227-262).  The two
 * P x V point regressors are CSR (int32 rowptr [P+1], int32 col, float64 val); vertices are
 * float64 [B,V,3] as in the reference.  align != 0 removes the mean offset of the regressed
 * points.  err_out [B,P] float64, err_mean_out [B] float64 (may be NULL). */
int shapy_p2p_error_f64(const int32_t *in_rowptr, const int32_t *in_col, const double *in_val,
                        const int32_t *tgt_rowptr, const int32_t *tgt_col, const double *tgt_val,
                        const double *input_verts, const double *target_verts, int B, int P,
                        int V_in, int V_tgt, int align, double *err_out, double *err_mean_out,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHAPY_HIP_H */
