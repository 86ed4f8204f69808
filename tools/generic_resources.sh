#!/bin/bash
# VGPR / SGPR / scratch (= spilled registers) of every kernel the generic registry (csrc/stencil_generic.hip) instantiates for a solution,
# from a device-only compile -- runs anywhere hipcc does, no GPU.  Counterpart of kernel_resources.sh for the hand-written registries.
# Solutions with box / plane neighbourhoods have a second translation unit without packed fp32 (csrc/Makefile np_hint): its kernels are
# listed too, as NoPk<part>.
# usage: tools/generic_resources.sh <solution>     e.g.  tools/generic_resources.sh tti | grep box_
S=${1:?solution name, e.g. cube}
cd "$(dirname "$0")/../yask_amd/csrc"
NS=$(sed -n 's/^namespace \(ykh_gen_[A-Za-z0-9_]*\) {.*/\1/p' gen/${S}_cdna4_hip.hpp | head -1)
HINT=$(awk '/ykh-build-hint: max-mixed-reads/ { if ($4 > 8) print 1 }' gen/${S}_cdna4_hip.hpp)
one() {   # extra flags
  T=$(mktemp)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. -mllvm -inline-threshold=1000000 "$@" -DYKH_GEN_HEADER="\"gen/${S}_cdna4_hip.hpp\"" -DYKH_GEN_NS=$NS \
        --cuda-device-only -S stencil_generic.hip -o $T 2>/dev/null || { echo "compile failed" >&2; exit 1; }
  grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size)|^\s+\.name:" $T | paste - - - - - |
    sed 's/\s\+/ /g; s/\.private_segment_fixed_size/scratch/; s/\.group_segment_fixed_size: [0-9]* //; s/\.vgpr_count/vgpr/; s/\.sgpr_count/sgpr/; s/ \.name: / /' |
    c++filt | sed 's/void ykh:://; s/ykh_gen_[a-z0-9_]*:://g; s/(ykh::PartArgs)//; s/ykh::SubPart<\([a-z_0-9]*\), \([0-9]*\)ull>/\1[mask \2]/g; s/ykh::NoPk<\([a-z_0-9]*\)>/NoPk<\1>/g; s/ykh::Lift2D<\([a-z_0-9]*\)>/Lift2D<\1>/g'
  rm -f $T
}
one ${HINT:+-DYKH_HAS_NOPK_TU}
[ -n "$HINT" ] && one -Xclang -target-feature -Xclang -packed-fp32-ops -DYKH_NOPK_TU
