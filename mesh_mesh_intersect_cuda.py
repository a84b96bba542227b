"""Import-name shim: ``import mesh_mesh_intersect_cuda`` keeps working.

The reference builds a pybind11/CUDA extension with this module name
(mesh-mesh-intersection/setup.py, src/mesh_mesh_intersect.cpp:59-64) and calls
``mesh_mesh_intersect_cuda.mesh_to_mesh_forward(query, target, max_collisions=...,
print_timings=...)``.  This module re-exports the gfx950 implementation under that name so
``body_measurements.py``-style callers run unchanged.
"""
from shapy_amd.measurements.mesh_mesh_intersection import mesh_to_mesh_forward  # noqa: F401
