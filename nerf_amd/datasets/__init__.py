from .blender import BlenderDataset  # noqa: F401
