#!/bin/bash
# Register / scratch / LDS table of the library's kernels (no GPU needed): device-only compile of one source with extra flags.
#   tools/kres.sh nms.hip [extra hipcc flags...]   ->  one line per obb:: kernel
SRC=${1:-nms.hip}; shift
D=$(cd "$(dirname "$0")/../yolov5_obb_amd/csrc" && pwd)
T=$(mktemp -d /tmp/kres.XXXXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -Wno-unused-result -Wno-unused-value -I"$D/../../include" -I"$D" --cuda-device-only "$@" -c "$D/$SRC" -o $T/dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/gfx.o || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/gfx.o | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed|group_segment_fixed|vgpr_spill|agpr_count" \
  | paste - - - - - - - | sed 's/ \+/ /g' | grep "_ZN3obb" \
  | sed -E 's/_ZN3obb[0-9]+//; s/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.vgpr_spill_count/spill/; s/\.(vgpr|sgpr|agpr)_count/\1/g; s/\.name: //' \
  | awk -F'\t' '{print $3, "|", $2, $4, $5, $6, $7, $1}' | cut -c1-170
[ -n "$KEEP" ] && cp $T/gfx.o "$KEEP"
rm -rf $T
