"""Switch a checkout of hukaixuan19970627/yolov5_obb to this package without editing it.

    import yolov5_obb_amd.dropin as dropin
    dropin.install()            # with the reference root on sys.path, BEFORE `import val` / `import train` / `import detect`
    import val                  # the reference's own script, now running the HIP hot path

What it does (INTEGRATION.md section 1 as code):
  * registers this package's mirrors under the reference's module names, so that the reference never looks for its
    compiled extensions (`utils.nms_rotated`, `utils.nms_rotated.nms_rotated_ext`, `DOTA_devkit.poly_nms_gpu.*`);
  * imports the reference's `utils.general`, `utils.loss`, `utils.rboxs_utils`, `models.yolo` and replaces, in those
    modules, the hot-path objects by this package's (`non_max_suppression_obb`, `ComputeLoss`, `Detect`, `rbox2poly`,
    `poly2hbb`) -- scripts imported afterwards bind the replacements (`from utils.general import non_max_suppression_obb`);
  * `uninstall()` puts everything back.
Nothing of the reference is copied or modified on disk.  GPU only, like the package.
"""
import importlib
import sys

_saved = []          # (module object or None for sys.modules entries, attribute / module name, old value)
_MISSING = object()


def _set_module(name, mod):
    _saved.append((None, name, sys.modules.get(name, _MISSING)))
    sys.modules[name] = mod


def _set_attr(mod, name, value):
    _saved.append((mod, name, getattr(mod, name, _MISSING)))
    setattr(mod, name, value)


def install(patch_loaded_scripts=True):
    """Idempotent.  Returns the list of (module, attribute) pairs that were replaced."""
    if _saved:
        return [(m.__name__ if m is not None else "sys.modules", n) for m, n, _ in _saved]
    from . import nms_rotated_ext
    from .DOTA_devkit import poly_nms_gpu
    from .DOTA_devkit.poly_nms_gpu import nms_wrapper, poly_nms, poly_overlaps
    from .models import yolo as my_yolo
    from .utils import general as my_general
    from .utils import loss as my_loss
    from .utils import nms_rotated as my_nms
    from .utils import rboxs_utils as my_rbox
    from .utils.nms_rotated import nms_rotated_wrapper
    # 1. the compiled extensions of the reference, by name
    _set_module("utils.nms_rotated", my_nms)
    _set_module("utils.nms_rotated.nms_rotated_wrapper", nms_rotated_wrapper)
    _set_module("utils.nms_rotated.nms_rotated_ext", nms_rotated_ext)
    _set_module("DOTA_devkit.poly_nms_gpu", poly_nms_gpu)
    _set_module("DOTA_devkit.poly_nms_gpu.poly_nms", poly_nms)
    _set_module("DOTA_devkit.poly_nms_gpu.poly_overlaps", poly_overlaps)
    _set_module("DOTA_devkit.poly_nms_gpu.nms_wrapper", nms_wrapper)
    for parent, child, mod in (("utils", "nms_rotated", my_nms), ("DOTA_devkit", "poly_nms_gpu", poly_nms_gpu)):
        try:                                              # `import utils.nms_rotated; utils.nms_rotated.obb_nms` needs the attribute
            _set_attr(importlib.import_module(parent), child, mod)
        except ImportError:
            pass                                          # (a checkout without the devkit)
    # 2. the Python hot path inside the reference's own modules
    general = importlib.import_module("utils.general")
    _set_attr(general, "non_max_suppression_obb", my_general.non_max_suppression_obb)
    _set_attr(general, "obb_nms", my_nms.obb_nms)
    rbox = importlib.import_module("utils.rboxs_utils")
    for name in ("rbox2poly", "poly2hbb"):
        _set_attr(rbox, name, getattr(my_rbox, name))
    loss = importlib.import_module("utils.loss")
    _set_attr(loss, "ComputeLoss", my_loss.ComputeLoss)
    yolo = importlib.import_module("models.yolo")
    _set_attr(yolo, "Detect", my_yolo.Detect)           # parse_model resolves layer names in models.yolo's namespace
    # 3. scripts that were imported before install(): rebind the names they copied
    if patch_loaded_scripts:
        for script in ("val", "detect", "train"):
            m = sys.modules.get(script)
            if m is None:
                continue
            for name, value in (("non_max_suppression_obb", my_general.non_max_suppression_obb), ("ComputeLoss", my_loss.ComputeLoss),
                                ("rbox2poly", my_rbox.rbox2poly), ("poly2hbb", my_rbox.poly2hbb)):
                if hasattr(m, name):
                    _set_attr(m, name, value)
    return [(m.__name__ if m is not None else "sys.modules", n) for m, n, _ in _saved]


def uninstall():
    while _saved:
        mod, name, old = _saved.pop()
        if mod is None:
            if old is _MISSING:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
        elif old is _MISSING:
            delattr(mod, name)
        else:
            setattr(mod, name, old)
