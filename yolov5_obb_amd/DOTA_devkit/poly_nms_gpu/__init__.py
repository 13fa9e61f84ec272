"""Mirror of DOTA_devkit/poly_nms_gpu/__init__.py:1-3 (the reference exports only poly_overlaps here)."""
from .poly_overlaps import poly_overlaps

__all__ = ['poly_overlaps', 'poly_nms']
