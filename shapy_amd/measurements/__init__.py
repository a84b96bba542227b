from .body_measurements import BodyMeasurements
from .mesh_mesh_intersection import (MeshMeshIntersection, MeshMeshIntersectionFunction,
                                     mesh_to_mesh_forward)
