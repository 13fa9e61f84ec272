"""Mirror of the reference package ``utils.nms_rotated`` (utils/nms_rotated/__init__.py:1-3).  The compiled module the
reference keeps inside this package (``utils/nms_rotated/nms_rotated_ext``) is reachable under the same name."""
from ... import nms_rotated_ext
from .nms_rotated_wrapper import obb_nms, poly_nms

__all__ = ["obb_nms", "poly_nms", "nms_rotated_ext"]
