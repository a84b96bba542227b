"""Minimal binary PLY writer (the reference exports with trimesh, demo.py:329-335)."""
import numpy as np


def write_ply(path, vertices, faces):
    vertices = np.asarray(vertices, dtype='<f4')
    faces = np.asarray(faces, dtype='<i4')
    header = ('ply\nformat binary_little_endian 1.0\n'
              f'element vertex {len(vertices)}\nproperty float x\nproperty float y\nproperty float z\n'
              f'element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n')
    rec = np.empty(len(faces), dtype=[('n', 'u1'), ('v', '<i4', (3,))])
    rec['n'] = 3
    rec['v'] = faces
    with open(path, 'wb') as f:
        f.write(header.encode('ascii'))
        f.write(vertices.tobytes())
        f.write(rec.tobytes())
