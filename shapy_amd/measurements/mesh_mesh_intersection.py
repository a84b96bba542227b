"""Mesh-mesh intersection operator on gfx950.

Drop-in for the reference's native extension and its wrappers:
  * ``mesh_to_mesh_forward``  <- mesh_mesh_intersect_cuda.mesh_to_mesh_forward
    (mesh-mesh-intersection/src/mesh_mesh_intersect.cpp:36-64)
  * ``MeshMeshIntersectionFunction`` / ``MeshMeshIntersection``
    (mesh-mesh-intersection/mesh_mesh_intersection/mesh_mesh_intersection.py:32-62)
"""
import torch
import torch.autograd as autograd
import torch.nn as nn

from .. import _lib


def mesh_to_mesh_forward(query_triangles, target_triangles, max_collisions=16,
                         print_timings=False):
    """query [B,Q,3,3], target [B,F,3,3] (contiguous device tensors, both float32 or both float64: the
    reference dispatches on the floating type, mesh_mesh_intersect_cuda_op.cu:996) ->
    [collision_faces int64 [B, Q*max_collisions] (-1 = empty),
     collision_bcs [B, Q*max_collisions, 2, 3] in the triangles' dtype].

    Errors are Python exceptions (the reference prints and calls exit(0) on any CUDA error,
    mesh_mesh_intersect_cuda_op.cu:76-86 -- deliberately not replicated)."""
    for name, t in (('query_triangles', query_triangles), ('target_triangles', target_triangles)):
        if not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA tensor')          # CHECK_CUDA (:20-22)
        if not t.is_contiguous():
            raise RuntimeError(f'{name} must be contiguous')             # CHECK_CONTIGUOUS (:23-24)
    dt = query_triangles.dtype
    if dt != target_triangles.dtype or dt not in (torch.float32, torch.float64):
        raise NotImplementedError('query and target triangles must both be float32 or both float64 '
                                  f'(got {query_triangles.dtype} / {target_triangles.dtype})')
    if query_triangles.dim() != 4 or target_triangles.dim() != 4:
        raise RuntimeError('expected [B,Q,3,3] and [B,F,3,3]')
    lib = _lib.load()
    B, Q = query_triangles.shape[:2]
    F = target_triangles.shape[1]
    dev = query_triangles.device
    faces = torch.empty(B, Q * max_collisions, dtype=torch.int64, device=dev)
    if dt == torch.float64:
        # the reference's double instantiation: brute-force scan for every Q (csrc/measure.hip)
        bcs = torch.empty(B, Q * max_collisions, 2, 3, dtype=torch.float64, device=dev)
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.shapy_mesh_to_mesh_f64(
            _lib.ptr(query_triangles), _lib.ptr(target_triangles), B, Q, F, max_collisions,
            _lib.ptr(faces), _lib.ptr(bcs), _lib.ptr(overflow), _lib.current_stream()),
            'shapy_mesh_to_mesh_f64')
        mesh_to_mesh_forward.last_overflow = overflow
        return [faces, bcs]
    bcs = torch.empty(B, Q * max_collisions, 2, 3, dtype=torch.float32, device=dev)
    nbytes = lib.shapy_mesh_to_mesh_workspace_bytes(B, Q, F, max_collisions)
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.shapy_mesh_to_mesh_f32(
        _lib.ptr(query_triangles), _lib.ptr(target_triangles), B, Q, F, max_collisions,
        _lib.ptr(faces), _lib.ptr(bcs), _lib.ptr(ws), ws.numel(), _lib.ptr(overflow),
        _lib.current_stream()), 'shapy_mesh_to_mesh_f32')
    mesh_to_mesh_forward.last_overflow = overflow
    return [faces, bcs]


class MeshMeshIntersectionFunction(autograd.Function):

    @staticmethod
    @torch.no_grad()
    def forward(ctx, query_triangles, target_triangles, print_timings=False, max_collisions=32,
                *args, **kwargs):
        faces, bcs = mesh_to_mesh_forward(query_triangles, target_triangles,
                                          print_timings=print_timings,
                                          max_collisions=max_collisions)
        ctx.mark_non_differentiable(faces, bcs)
        return faces, bcs

    @staticmethod
    def backward(ctx, grad_output, *args, **kwargs):
        raise NotImplementedError


class MeshMeshIntersection(nn.Module):

    def __init__(self, max_collisions=32):
        super().__init__()
        self.max_collisions = max_collisions

    def forward(self, query_triangles, target_triangles, print_timings=False):
        return MeshMeshIntersectionFunction.apply(query_triangles, target_triangles,
                                                  print_timings, self.max_collisions)
