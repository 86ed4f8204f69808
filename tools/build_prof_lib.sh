#!/bin/bash
# Build libyask_kernel.<stencil>.cdna4_hip.so WITH the sweep / experiment shapes (-DYKH_PROFILING) into yask_amd/lib_prof/ (objects in
# yask_amd/csrc/build_prof/), next to the shipped library.  Use: YASK_HIP_LIB_DIR=$PWD/yask_amd/lib_prof python tools/...
#   tools/build_prof_lib.sh 3axis [iso3dfd ssg ...]
set -e
cd "$(dirname "$0")/../yask_amd/csrc"
make -j16 >/dev/null            # the runtime objects (build/*.o) the profiling library links with
mkdir -p build_prof ../lib_prof
for s in "$@"; do
  objs=""
  for f in stencil_$s.hip stencil_${s}_k*.hip; do
    [ -f "$f" ] || continue
    o=build_prof/${f%.hip}.o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer "$o" | head -1)" ]; then
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wno-unused-result -DYKH_PROFILING -c "$f" -o "$o" &
    fi
    objs="$objs $o"
  done
  wait
  rt=$(ls build/ykh_*.o) ; api=$(ls ../cxxapi/_build/yk_hip_adapter.o ../cxxapi/_build/com_*.o ../cxxapi/_build/fd_coeff.o 2>/dev/null || true)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_prof/libyask_kernel.$s.cdna4_hip.so $objs $rt $api -ldl -lpthread
  echo "built yask_amd/lib_prof/libyask_kernel.$s.cdna4_hip.so"
done
