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
Nothing of the reference is copied or modified on disk.

CPU tensors (BASELINE configs[0]: `detect.py --device cpu`).  The package itself has no CPU path and never falls back to
one.  The reference dispatches on the tensor's device (nms_rotated_ext.cpp:25-39: CUDA -> nms_rotated_cuda, else
nms_rotated_cpu), so the objects install() binds do the same one level up: a GPU tensor goes to this package, a CPU tensor
to the REFERENCE'S OWN object that install() displaced (its `non_max_suppression_obb`, `obb_nms` with its compiled CPU
extension, `ComputeLoss`, `Detect.forward`) -- the reference's code, unchanged, exactly as if install() had not been called.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys

_saved = []          # (module object or None for sys.modules entries, attribute / module name, old value)
_MISSING = object()


def _set_module(name, mod):
    _saved.append((None, name, sys.modules.get(name, _MISSING)))
    sys.modules[name] = mod


def _set_attr(mod, name, value):
    _saved.append((mod, name, getattr(mod, name, _MISSING)))
    setattr(mod, name, value)


_ref_nms_pkg = {}    # the reference's own utils.nms_rotated package, imported on first CPU use under a private name


def _reference_obb_nms(ref_root, cpu_ext):
    """The reference's obb_nms (utils/nms_rotated/nms_rotated_wrapper.py:6-46) with ITS compiled CPU extension, imported
    from the reference tree under a private package name (`utils.nms_rotated` itself now names this package's mirror)."""
    if "mod" not in _ref_nms_pkg:
        pkg_dir = os.path.join(ref_root, "utils", "nms_rotated")
        name = "_reference_utils_nms_rotated"
        if cpu_ext:                                       # a build of the reference's extension that lives outside its tree
            loader = importlib.machinery.ExtensionFileLoader("nms_rotated_ext", cpu_ext)
            spec_e = importlib.util.spec_from_file_location("nms_rotated_ext", cpu_ext, loader=loader)
            ext = importlib.util.module_from_spec(spec_e)
            loader.exec_module(ext)
            sys.modules[name + ".nms_rotated_ext"] = ext
        spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        try:
            spec.loader.exec_module(mod)
        except ImportError as e:
            sys.modules.pop(name, None)
            raise RuntimeError("CPU tensors run the reference's own nms_rotated_cpu: build its extension first (python setup.py "
                               "build_ext --inplace in utils/nms_rotated of the reference) -- yolov5_obb_amd has no CPU path") from e
        _ref_nms_pkg["mod"] = mod
    return _ref_nms_pkg["mod"].obb_nms


def _is_cuda(x):
    import torch
    return isinstance(x, torch.Tensor) and x.is_cuda


def install(patch_loaded_scripts=True, reference_cpu_ext=None):
    """Idempotent.  Returns the list of (module, attribute) pairs that were replaced.
    reference_cpu_ext: optional path of a built `nms_rotated_ext*.so` of the reference (when it was not built in-tree)."""
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
    # 2. the Python hot path inside the reference's own modules; every callable dispatches on the device like the
    #    reference's extension does (nms_rotated_ext.cpp:25-39): GPU -> this package, CPU -> the reference's own object
    general = importlib.import_module("utils.general")
    ref_root = os.path.dirname(os.path.dirname(os.path.abspath(general.__file__)))
    ref_nmsobb = general.non_max_suppression_obb

    def non_max_suppression_obb(prediction, *args, **kwargs):
        return (my_general.non_max_suppression_obb if _is_cuda(prediction) else ref_nmsobb)(prediction, *args, **kwargs)

    def obb_nms(dets, scores, iou_thr, device_id=None):
        on_gpu = _is_cuda(dets) or (not hasattr(dets, "is_cuda") and device_id is not None)      # ndarray + device_id -> cuda:{id}
        return (my_nms.obb_nms if on_gpu else _reference_obb_nms(ref_root, reference_cpu_ext))(dets, scores, iou_thr, device_id)

    non_max_suppression_obb.__doc__ = my_general.non_max_suppression_obb.__doc__
    non_max_suppression_obb.hip = my_general.non_max_suppression_obb
    obb_nms.__doc__ = my_nms.obb_nms.__doc__
    obb_nms.hip = my_nms.obb_nms
    _set_attr(general, "non_max_suppression_obb", non_max_suppression_obb)
    _set_attr(general, "obb_nms", obb_nms)
    rbox = importlib.import_module("utils.rboxs_utils")
    for name in ("rbox2poly", "poly2hbb"):
        _set_attr(rbox, name, getattr(my_rbox, name))   # (both keep the reference's host code for CPU tensors / ndarrays)
    loss = importlib.import_module("utils.loss")
    ref_loss = loss.ComputeLoss

    def ComputeLoss(model, autobalance=False):
        """utils/loss.py:91-120: the loss object of a model -- HIP kernels for a model on the GPU, the reference's class otherwise."""
        dev = next(model.parameters()).device
        return (my_loss.ComputeLoss if dev.type == "cuda" else ref_loss)(model, autobalance)

    ComputeLoss.hip = my_loss.ComputeLoss
    _set_attr(loss, "ComputeLoss", ComputeLoss)
    yolo = importlib.import_module("models.yolo")
    ref_detect = yolo.Detect
    _set_attr(my_yolo.Detect, "_cpu_forward", ref_detect.forward)   # eval-mode forward of CPU tensors: the reference's own code
    # checkpoints pickle the class by module path: written under install() they name `models.yolo.Detect`, which a plain
    # checkout of the reference resolves to its own class (same constructor, attributes and parameter names)
    _set_attr(my_yolo.Detect, "__module__", "models.yolo")
    _set_attr(yolo, "Detect", my_yolo.Detect)           # parse_model resolves layer names in models.yolo's namespace
    # 3. scripts that were imported before install(): rebind the names they copied
    if patch_loaded_scripts:
        for script in ("val", "detect", "train"):
            m = sys.modules.get(script)
            if m is None:
                continue
            for name, value in (("non_max_suppression_obb", non_max_suppression_obb), ("ComputeLoss", ComputeLoss),
                                ("rbox2poly", my_rbox.rbox2poly), ("poly2hbb", my_rbox.poly2hbb)):
                if hasattr(m, name):
                    _set_attr(m, name, value)
    return [(m.__name__ if m is not None else "sys.modules", n) for m, n, _ in _saved]


def uninstall():
    for k in list(_ref_nms_pkg):
        _ref_nms_pkg.pop(k)
    for name in ("_reference_utils_nms_rotated", "_reference_utils_nms_rotated.nms_rotated_wrapper", "_reference_utils_nms_rotated.nms_rotated_ext"):
        sys.modules.pop(name, None)
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
