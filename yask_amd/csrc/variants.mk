# Compile-time variants of solutions: the same DSL definition compiled with other compiler options -- the cases of the reference's
# own test matrix that set `radius=` / `domain_dims=` (src/kernel/Makefile:1116-1153: 2d-tests, 3d-tests, 3d-tests3).  A variant is
# built like the reference builds it: one more library, named <stencil><suffix> (the reference's YK_STENCIL_SUFFIX,
# src/kernel/Makefile:203-216: libyask_kernel.<stencil><suffix>.<arch>.so).  Included by ../compiler/Makefile (renders
# gen/<tag>_cdna4_hip.hpp) and by this directory's Makefile (builds lib/libyask_kernel.<tag>.cdna4_hip.so); oracle/Makefile takes
# the same flags as YC_EXTRA for the reference build the goldens come from (tests/golden/make_golden.py VARIANT_CASES).
#   VS_<tag> = the solution's registered name, VF_<tag> = the extra yask_compiler flags
VARIANTS := iso3dfd-r3zxy iso3dfd_sponge-r6 test_stream_3d-r5 test_3d-zyx test_stages_3d-xzy test_partial_3d-xzy test_2d-yx test_reverse_2d-r1
VS_iso3dfd-r3zxy       := iso3dfd
VF_iso3dfd-r3zxy       := -radius 3 -domain-dims z,x,y
VS_iso3dfd_sponge-r6   := iso3dfd_sponge
VF_iso3dfd_sponge-r6   := -radius 6
VS_test_stream_3d-r5   := test_stream_3d
VF_test_stream_3d-r5   := -radius 5
VS_test_3d-zyx         := test_3d
VF_test_3d-zyx         := -domain-dims z,y,x
VS_test_stages_3d-xzy  := test_stages_3d
VF_test_stages_3d-xzy  := -domain-dims x,z,y
VS_test_partial_3d-xzy := test_partial_3d
VF_test_partial_3d-xzy := -domain-dims x,z,y
VS_test_2d-yx          := test_2d
VF_test_2d-yx          := -domain-dims y,x
VS_test_reverse_2d-r1  := test_reverse_2d
VF_test_reverse_2d-r1  := -radius 1
