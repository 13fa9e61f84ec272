#!/bin/bash
# Per-workgroup timeline of k_nms_small on the headline step: a library built with -DOBB_SMALL_TRACE (in-kernel printf of wall-clock
# stamps), four calls on one bench tensor, the last call's lines analysed.  Build here (no GPU needed): tools/small_trace.sh build
# then on the GPU box: tools/small_trace.sh run <outdir>
D=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  cd $D/yolov5_obb_amd/csrc && rm -rf /tmp/trace_objs; mkdir -p /tmp/trace_objs && for f in *.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -Wno-unused-result -Wno-unused-value -Wno-format -DOBB_SMALL_TRACE $TRACE_EXTRA -I../../include -I. -c $f -o /tmp/trace_objs/${f%.hip}.o & done; wait
  rm -f $D/yolov5_obb_amd/libobb_hip_trace.so; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/yolov5_obb_amd/libobb_hip_trace.so /tmp/trace_objs/*.o && nm -D $D/yolov5_obb_amd/libobb_hip_trace.so | grep -q obb_debug_small_trace && echo built
  exit
fi
O=${2:-gpurun_out/trace}; mkdir -p $O
for h in ${HELPERS:-0 256}; do
OBB_NMS_SMALL_HELPERS=$h OBB_HIP_LIB=$D/yolov5_obb_amd/libobb_hip_trace.so OBB_BINDING=${BINDING:-ctypes} python tools/small_trace_report.py 2>&1 | grep -v amdgpu.ids | tee $O/report_h$h.txt
done
