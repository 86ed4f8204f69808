#!/bin/bash
# One line per translation unit of the hot-path kernel libraries: sha256 of the gfx950 assembly hipcc emits for it (device-only, -S).
# Purpose: PROVE that an edit of a shared header (ykh_starlin.hpp, ykh_march.hpp, ykh_device.hpp ...) left the kernels that already
# ship untouched -- same assembly, same behaviour on the GPU -- when no GPU is at hand to re-run the suite:
#     tools/isa_fingerprint.sh > /tmp/after.txt ; diff profiles/isa_fingerprints.txt /tmp/after.txt
# (a new template flag behind `if constexpr` must not change a single instruction of the instantiations that do not use it).
# usage: tools/isa_fingerprint.sh [stencil ...]      (default: iso3dfd 3axis 3axis_r1 ssg); runs anywhere hipcc does.
cd "$(dirname "$0")/../yask_amd/csrc"
S=${@:-iso3dfd 3axis 3axis_r1 ssg}
T=$(mktemp -d)
for s in $S; do
  for f in stencil_${s}.hip stencil_${s}_k*.hip; do
    [ -f "$f" ] || continue
    ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. --cuda-device-only -S "$f" -o "$T/$f.s" 2>/dev/null &&
      # (the assembly carries no paths or time stamps; .ident names the compiler build, __hip_cuid_<random> is a per-compilation id)
      echo "$f $(grep -v '^\s*\.ident' "$T/$f.s" | sed 's/__hip_cuid_[0-9a-f]*/__hip_cuid_X/g' | sha256sum | cut -d' ' -f1) $(grep -c '^\s*\.amdhsa_kernel ' "$T/$f.s") kernels" > "$T/$f.txt" ) &
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.2; done
  done
done
wait
cat "$T"/*.txt | sort
rm -rf "$T"
