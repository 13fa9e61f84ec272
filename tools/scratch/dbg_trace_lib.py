"""Development aid: run the configs[1] step on a traced build of the library.

    cd yolov5_obb_amd/csrc && hipcc <CXXFLAGS of the Makefile> -DOBB_SORT_TRACE -c nms.hip -o /tmp/nms_trace.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libobb_trace.so /tmp/nms_trace.o $(ls *.o | grep -v '^nms.o')

OBB_SORT_TRACE makes workgroup 0 of k_sort_prep_lds print its phase times (load / sort / write+segments / records)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov5_obb_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libobb_trace.so")
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 15, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    out = non_max_suppression_obb(pred, **kw)
torch.cuda.synchronize()
