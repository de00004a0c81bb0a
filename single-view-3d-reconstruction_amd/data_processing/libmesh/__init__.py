from .inside_mesh import MeshIntersector, check_mesh_contains  # noqa: F401
