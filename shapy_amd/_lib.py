"""ctypes binding of libshapy_hip.so (the C-ABI declared in include/shapy_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load,
every operator raises ``ShapyHipError``.
"""
import ctypes
import os
import os.path as osp

_HERE = osp.dirname(osp.abspath(__file__))
# SHAPY_HIP_LIB selects a variant build of the same library (kernel tuning experiments only)
LIB_PATH = os.environ.get('SHAPY_HIP_LIB') or osp.join(_HERE, 'csrc', 'libshapy_hip.so')

c_float_p = ctypes.POINTER(ctypes.c_float)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
vp = ctypes.c_void_p
i32 = ctypes.c_int32
i64 = ctypes.c_int64


class ShapyHipError(RuntimeError):
    pass


class ShapyConv(ctypes.Structure):
    _fields_ = [('in_', vp), ('wgt', vp), ('bias', vp), ('res', vp), ('out', vp),
                ('B', i32), ('Hi', i32), ('Wi', i32), ('Cin', i32), ('in_ld', i32),
                ('Ho', i32), ('Wo', i32), ('Cout', i32),
                ('ksize', i32), ('stride', i32), ('pad', i32),
                ('out_ld', i32), ('out_coff', i32), ('res_ld', i32), ('res_coff', i32),
                ('relu', i32), ('ups', i32), ('tile', i32), ('dtype', i32),
                ('split_kib', i32), ('wgt_wino', vp), ('split_ws', vp), ('split_cnt', vp), ('split_cnt_n', i32)]


class ShapyOp(ctypes.Structure):
    _fields_ = [('type', i32), ('lane', i32), ('barrier_before', i32),
                ('Hi', i32), ('Wi', i32), ('Cin', i32), ('in_ld', i32), ('Ho', i32), ('Wo', i32),
                ('Cout', i32), ('ksize', i32), ('stride', i32), ('pad', i32),
                ('out_ld', i32), ('out_coff', i32), ('res_ld', i32), ('res_coff', i32),
                ('relu', i32), ('ups', i32), ('tile', i32), ('group', i32),
                ('sig', i32), ('wait', i32 * 3),
                ('in_off', i64), ('out_off', i64), ('res_off', i64),
                ('wgt_off', i64), ('bias_off', i64), ('wino_off', i64),
                ('split_off', i64), ('cnt_off', i64), ('split_floats', i64), ('cnt_n', i64)]


class ShapySmplxModel(ctypes.Structure):
    _fields_ = [('V', i32), ('J', i32), ('NB', i32), ('P', i32), ('Ppad', i32), ('NBpad', i32),
                ('n_static_lmk', i32), ('n_dyn_lmk', i32), ('n_dyn_rows', i32), ('n_neck', i32),
                ('parents', vp), ('J_template', vp), ('J_shapedirs', vp), ('v_template', vp),
                ('shapedirs_t', vp), ('posedirs_t', vp), ('lbs_weights_t', vp), ('faces', vp),
                ('lmk_faces_idx', vp), ('lmk_bary', vp), ('dyn_lmk_faces_idx', vp),
                ('dyn_lmk_bary', vp), ('neck_kin_chain', vp)]


OP_CONV, OP_STEM, OP_MEANPOOL = 0, 1, 2
POSE_ROTMAT, POSE_CONT6D, POSE_AXIS_ANGLE = 0, 1, 2
DTYPE_F32, DTYPE_BF16, DTYPE_F32X6 = 0, 1, 2
TILES = {'auto': 0, '256x48': 1, '128x96': 2, '128x128': 3, '256x64': 4, '64x48': 5,
         '64x96': 6, '64x128': 7, '64x64': 8, '128x48': 9, '128x64': 10, '32x64': 11}
for _k, _v in list(TILES.items()):      # tuning knobs (csrc/conv_igemm.hip: conv2d)
    TILES[_k + '+noswz'] = _v | 0x400
    TILES[_k + '+bk32'] = _v | 0x200
    TILES[_k + '+bk16'] = _v | 0x800
    TILES[_k + '+noswz+bk16'] = _v | 0xC00
    TILES[_k + '+direct'] = _v | 0x2000       # never take the Winograd path
    TILES[_k + '+wtm1'] = _v | 0x4000         # Winograd: 16 tiles per workgroup
    TILES[_k + '+wtm2'] = _v | 0x8000         # Winograd: 32 tiles per workgroup
    TILES[_k + '+nonslab'] = _v | 0x10000     # keep the m-major XCD order for large weights
    TILES[_k + '+pd3'] = _v | 0x40000         # implicit GEMM: 3 chunks of loads in flight (bf16 default)
    TILES[_k + '+pd1'] = _v | 0x80000         # ... 1 chunk (float32 default)
    TILES[_k + '+noallk'] = _v | 0x20000      # Winograd: K loop chunk by chunk even for Cin = 48 / 64
TILE_X6 = 0x800000                      # in a float32 op list: this layer on the bf16x6 kernel (split weight planes)
TILE_WINO4 = 0x100000                   # ShapyConv.wgt_wino holds F(4x4,3x3) filters (conv_wino4.hip)


def tile_w4_ksplit(s):
    """SHAPY_TILE_KSPLIT(s): a layer with its K loop cut into s = 1..4 slices (bits 21..22)."""
    if not 1 <= int(s) <= 4:
        raise ValueError(f'F(4x4) split-K: 1..4 slices, got {s}')
    return (int(s) - 1) << 21


def igemm_split_sizes(Ho, Wo, cout, s):
    """(slab floats per image, counter ints per image) that cover an implicit-GEMM split-K layer with Ho x Wo
    output pixels per image whatever tile (<= 64 x 64) the library picks (include/shapy_hip.h)."""
    hw = Ho * Wo
    return s * (hw + 63) * (cout + 63), 8 * (hw // 32 + 1) * (cout // 48 + 1)


def w4_split_sizes(H, W, cout, s):
    """(slab floats per image, counter ints per image) of an F(4x4) split-K layer on an H x W map
    (include/shapy_hip.h: ShapyConv.split_ws / split_cnt)."""
    t = ((H + 3) // 4) * ((W + 3) // 4)
    return s * t * 16 * cout, 2 * ((t + 15) // 16) * (cout // 16)


#: every symbol include/shapy_hip.h declares: (restype, argtypes)
SIGNATURES = {
    'shapy_abi_version': (ctypes.c_int, []),
    'shapy_build_arch': (ctypes.c_char_p, []),
    'shapy_conv2d': (ctypes.c_int, [ctypes.POINTER(ShapyConv), vp]),
    'shapy_conv2d_group': (ctypes.c_int, [ctypes.POINTER(ShapyConv), ctypes.c_int, vp]),
    'shapy_hrnet_run': (ctypes.c_int, [ctypes.POINTER(ShapyOp), ctypes.c_int, vp, vp, vp, i64, vp, i64, vp,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, vp]),
    'shapy_hrnet_graph_create': (ctypes.c_int, [ctypes.POINTER(ShapyOp), ctypes.c_int, vp, vp, vp, i64,
                                                vp, i64, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_void_p)]),
    'shapy_hrnet_lane_stream': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
    'shapy_hrnet_graph_launch': (ctypes.c_int, [vp, vp]),
    'shapy_hrnet_graph_destroy': (ctypes.c_int, [vp]),
    'shapy_regressor_affine_f32': (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, vp]),
    'shapy_regressor_collapsed_f32': (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_int, vp]),
    'shapy_joint_regress_f32': (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, vp]),
    'shapy_smplx_pose_f32': (ctypes.c_int, [ctypes.POINTER(ShapySmplxModel), vp, ctypes.c_int,
                                            ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp]),
    'shapy_head_prepare_f32': (ctypes.c_int, [vp] + [ctypes.c_int] * 10 + [vp, vp, vp, vp]),
    'shapy_pose_decode_f32': (ctypes.c_int, [vp, ctypes.c_int, vp, i64, vp]),
    'shapy_weak_persp_project_f32': (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, vp]),
    'shapy_smplx_skin_f32': (ctypes.c_int, [ctypes.POINTER(ShapySmplxModel), vp, vp, vp,
                                            ctypes.c_int, vp]),
    'shapy_smplx_joints_f32': (ctypes.c_int, [ctypes.POINTER(ShapySmplxModel), vp, vp, vp, vp, vp,
                                              vp, vp, ctypes.c_int, ctypes.c_int, vp]),
    'shapy_smplx_prepare_f32': (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32),
                                               ctypes.POINTER(ctypes.c_int64), ctypes.c_int, vp, i64, ctypes.c_int,
                                               vp, i64, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]),
    'shapy_smplx_forward_f32': (ctypes.c_int, [ctypes.POINTER(ShapySmplxModel), vp, ctypes.c_int,
                                               ctypes.c_int] + [vp] * 15 +
                                [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'shapy_b2a_polynomial_f32': (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, vp]),
    'shapy_crop_resize_normalize_u8': (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                                      c_float_p, c_float_p, vp]),
    'shapy_aligned_point_error_f32': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     vp, vp, vp, vp]),
    'shapy_p2p_error_f64': (ctypes.c_int, [vp] * 8 + [ctypes.c_int] * 5 + [vp, vp, vp]),
    'shapy_mesh_to_mesh_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'shapy_mesh_to_mesh_f32': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, vp, vp, vp, ctypes.c_size_t, vp, vp]),
    'shapy_mesh_to_mesh_f64': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, vp, vp, vp, vp]),
    'shapy_body_measure_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'shapy_body_measure_f32': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              c_i32_p, c_float_p, ctypes.c_int, vp, vp,
                                              ctypes.c_size_t, vp, vp]),
}

_lib = None


def load(build_if_missing=False):
    """Loads the shared library and binds every declared symbol.  Never builds implicitly
    on the product path; ``__graft_entry__.build()`` / ``python -m shapy_amd.build`` do."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: libshapy_hip.so has to bind to the SAME HIP runtime
    # (libamdhip64) instance torch uses, otherwise its streams / device pointers are foreign
    # to our launches (observed: hipErrorNoDevice when our library was loaded first).
    import torch  # noqa: F401
    if not osp.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _build
            _build.build()
        else:
            raise ShapyHipError(
                f'{LIB_PATH} not found: build it with `python -m shapy_amd.build` '
                '(there is no CPU fallback)')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise ShapyHipError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ShapyHipError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise ShapyHipError(f'{what} failed with code {rc}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name):
    if not t.is_cuda:
        raise ShapyHipError(f'{name} must live on the GPU (got {t.device}); the HIP path has no '
                            'CPU fallback')
