"""Mirror of the reference package ``utils.nms_rotated`` (utils/nms_rotated/__init__.py:1-3)."""
from .nms_rotated_wrapper import obb_nms, poly_nms

__all__ = ["obb_nms", "poly_nms"]
