// ykh_fn.hpp -- device math functions the generated parts may call (included by gen/*.hpp).
#pragma once
#include <hip/hip_runtime.h>

namespace ykh {

// ------------------------------------------------------------------ math functions of the YASK DSL
// (yc_node_factory::new_math_func nodes, include/aux/yc_node_api.hpp), applied element-wise to z-vectors
#define YKH_FN1(name, expr)                                                                          \
    __device__ __forceinline__ float fn_##name(float x) { return expr; }                             \
    __device__ __forceinline__ double fn_##name(double x) { return expr; }                           \
    template <typename V, typename = decltype(V()[0])>                                               \
    __device__ __forceinline__ V fn_##name(V v) {                                                    \
        V r;                                                                                         \
        for (int i = 0; i < (int)(sizeof(V) / sizeof(v[0])); i++) r[i] = fn_##name(v[i]);            \
        return r;                                                                                    \
    }
YKH_FN1(sqrt, ::sqrt(x)) YKH_FN1(cbrt, ::cbrt(x)) YKH_FN1(fabs, ::fabs(x)) YKH_FN1(erf, ::erf(x)) YKH_FN1(exp, ::exp(x))
YKH_FN1(log, ::log(x)) YKH_FN1(sin, ::sin(x)) YKH_FN1(cos, ::cos(x)) YKH_FN1(atan, ::atan(x))
#undef YKH_FN1
#define YKH_FN2(name, expr)                                                                          \
    __device__ __forceinline__ float fn_##name(float x, float y) { return expr; }                    \
    __device__ __forceinline__ double fn_##name(double x, double y) { return expr; }                 \
    template <typename V, typename = decltype(V()[0])>                                               \
    __device__ __forceinline__ V fn_##name(V a, V b) {                                               \
        V r;                                                                                         \
        for (int i = 0; i < (int)(sizeof(V) / sizeof(a[0])); i++) r[i] = fn_##name(a[i], b[i]);      \
        return r;                                                                                    \
    }
YKH_FN2(pow, ::pow(x, y)) YKH_FN2(max, (x > y ? x : y)) YKH_FN2(min, (x < y ? x : y))
#undef YKH_FN2

}  // namespace ykh
