#!/bin/bash
# Build generic-registry libraries WITH the experimental marching shapes (-DYKH_MARCH_EXP, stencil_generic.hip) into yask_amd/lib_x/
# (objects in yask_amd/csrc/build_x/), next to the shipped ones.  Use: YASK_HIP_LIB_DIR=$PWD/yask_amd/lib_x python tools/...
#   tools/build_exp_lib.sh awp awp_elastic ...        (the runtime objects csrc/build/*.o must be current: make -C yask_amd/csrc)
set -e
cd "$(dirname "$0")/../yask_amd/csrc"
mkdir -p build_x ../lib_x
for s in "$@"; do
  NS=$(sed -n 's/^namespace \(ykh_gen_[A-Za-z0-9_]*\) {.*/\1/p' gen/${s}_cdna4_hip.hpp | head -1)
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wno-unused-result -mllvm -inline-threshold=1000000 -DYKH_MARCH_EXP ${EXP_FLAGS} \
        -DYKH_GEN_HEADER="\"gen/${s}_cdna4_hip.hpp\"" -DYKH_GEN_NS=$NS -c stencil_generic.hip -o build_x/stencil_generic_$s.o &
done
wait
for s in "$@"; do
  rt=$(ls build/ykh_*.o) ; api=$(ls ../cxxapi/_build/yk_hip_adapter.o ../cxxapi/_build/com_*.o ../cxxapi/_build/fd_coeff.o 2>/dev/null || true)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_x/libyask_kernel.$s.cdna4_hip.so build_x/stencil_generic_$s.o $rt $api -ldl -lpthread
  echo "built yask_amd/lib_x/libyask_kernel.$s.cdna4_hip.so"
done
