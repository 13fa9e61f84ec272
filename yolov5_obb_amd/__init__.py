"""yolov5_obb_amd -- MI355X (gfx950) native oriented-box hot path for yolov5_obb.

The package mirrors the reference's module names for the hot path only
(``utils.nms_rotated``, ``nms_rotated_ext``, ``utils.general.non_max_suppression_obb``,
``utils.rboxs_utils``, ``utils.loss.ComputeLoss``, ``models.yolo.Detect``,
``DOTA_devkit.poly_nms_gpu``); every public function cites the reference
file:line it replaces.  All compute goes through the C ABI of
``libobb_hip.so`` (``include/obb_hip.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"
