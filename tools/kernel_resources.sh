#!/bin/bash
# Print VGPR/SGPR/spill/LDS usage of every kernel in the stencil translation units (device-only compile).
# usage: tools/kernel_resources.sh [stencil]    (default iso3dfd)
S=${1:-iso3dfd}
cd "$(dirname "$0")/../yask_amd/csrc"
for f in stencil_${S}_k*.hip; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. --cuda-device-only -S $f -o /tmp/kr_$$_$f.s 2>/dev/null
    grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size)|^\s+\.name:" /tmp/kr_$$_$f.s | paste - - - - - |
      sed 's/\s\+/ /g; s/\.private_segment_fixed_size/scratch/; s/\.vgpr_spill_count/spill/; s/\.vgpr_count/vgpr/; s/\.sgpr_count/sgpr/; s/ \.name: / /' > /tmp/kr_$$_$f.txt
    rm -f /tmp/kr_$$_$f.s ) &
done
wait
cat /tmp/kr_$$_*.txt | c++filt | sed 's/void ykh:://; s/ykh_gen_[a-z0-9_]*:://; s/(ykh::PartArgs)//'
rm -f /tmp/kr_$$_*.txt
