/* ORACLE (test infrastructure only): the float64 instantiation of oracle/mesh_intersect.c -- the reference
 * dispatches mesh_to_mesh_forward on the floating type (mesh_mesh_intersect_cuda_op.cu:996).  Same source,
 * REAL = double, symbols shapy_oracle_*_f64. */
#define REAL double
#define SFX(name) name##_f64
#define RMIN fmin
#define RMAX fmax
#include "mesh_intersect.c"
