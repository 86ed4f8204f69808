/* oracle/stencil_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker; never the product path).
 *
 * A plain-C, single-threaded (optionally OpenMP over x) restatement of the arithmetic of the three
 * intel/yask stencils on the hot path, written from the reference's DSL definitions and its
 * generated scalar code. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Parity status: PINNED -- tests/test_oracle_vs_reference.py checks every
 * function here against outputs of the unmodified reference (oracle/_ref, built by oracle/Makefile)
 * committed as fixtures under tests/golden/.
 *
 * Reference sources restated (paths relative to /root/reference):
 *   iso3dfd : src/stencils/Iso3dfdStencil.cpp:63-137  (expression + delta_xyz=50, centre x3)
 *   3axis   : src/stencils/SimpleStencils.cpp:61-103  (average of 6R+1 points)
 *   ssg     : src/stencils/SSGElasticStencil.cpp:98-189, src/stencils/ElasticStencil/ElasticStencil.hpp:86-300
 *   weights : src/contrib/coefficients/fd_coeff.cpp:54-101 (Fornberg), src/common/fd_coeff2.cpp:49-57
 *   numeric form: generated calc_scalar (emitter src/compiler/lib/YaskKernel.cpp:561-589): every
 *             named temporary is a real_t, constants are double literals, so a product with a
 *             constant is evaluated in double and rounded when stored to the temporary.
 *   step slots: imod_flr(t, nslots) (src/kernel/lib/yk_var.hpp:131-147); in-place update of the
 *             oldest slot (src/compiler/lib/Var.cpp:435-465).
 *   boundary: single rank => halo values are never written by the solver; they keep their initial
 *             values and are read as boundary data (src/kernel/lib/halo.cpp:84, context.cpp:249-256).
 *
 * Array layout used by this oracle (its own, NOT the reference's folded layout): a box with a
 * uniform halo H on every side, row-major [x][y][z], z fastest:
 *     idx(x,y,z) = ((x+H)*(ny+2H) + (y+H))*(nz+2H) + (z+H),   -H <= x < nx+H etc.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define YO_EXPORT __attribute__((visibility("default")))

/* ---- logical-index hash: value in [-1,1) as a function of (var id, step slot, x, y, z) ---- */
static inline double yo_hash_unit_i(int64_t vid, int64_t slot, int64_t x, int64_t y, int64_t z) {
    uint32_t u = (uint32_t)x * 0x9E3779B1u ^ (uint32_t)y * 0x85EBCA77u ^ (uint32_t)z * 0xC2B2AE3Du ^
                 (uint32_t)slot * 0x27D4EB2Fu ^ (uint32_t)vid * 0x165667B1u;
    u ^= u >> 15; u *= 0x2C1B3C6Du; u ^= u >> 12; u *= 0x297A2D39u; u ^= u >> 15;
    return (double)(int32_t)u * (1.0 / 2147483648.0);
}
YO_EXPORT double yo_hash_unit(int64_t vid, int64_t slot, int64_t x, int64_t y, int64_t z) {
    return yo_hash_unit_i(vid, slot, x, y, z);
}

/* Fornberg weights for derivative order `order` at point `eval_point` on grid `pts[0..n)`.
 * Restates src/contrib/coefficients/fd_coeff.cpp:54-101. */
YO_EXPORT void yo_fd_coeff(double* coeff, double eval_point, int order, const double* pts, int n) {
    double c1, c2, c3, x_0 = eval_point;
    int m = order + 1;
    double* d = (double*)calloc((size_t)m * n * n, sizeof(double));
#define D(k, i, j) d[((size_t)(k) * n + (i)) * n + (j)]
    D(0, 0, 0) = 1.0;
    c1 = 1.0;
    for (int nn = 1; nn <= n - 1; ++nn) {
        c2 = 1.0;
        for (int v = 0; v < nn; ++v) {
            c3 = pts[nn] - pts[v];
            c2 = c2 * c3;
            for (int k = 0; k <= (nn < order ? nn : order); ++k) {
                D(k, nn, v) = ((pts[nn] - x_0) * D(k, nn - 1, v) - (k > 0 ? k * D(k - 1, nn - 1, v) : 0.0)) / c3;
            }
        }
        for (int k = 0; k <= (nn < order ? nn : order); ++k) {
            D(k, nn, nn) = (c1 / c2) * ((k > 0 ? k * D(k - 1, nn - 1, nn - 1) : 0.0) - (pts[nn - 1] - x_0) * D(k, nn - 1, nn - 1));
        }
        c1 = c2;
    }
    for (int i = 0; i < n; ++i) coeff[i] = D(order, n - 1, i);
#undef D
    free(d);
}

/* get_center_fd_coefficients(order, radius): 2*radius+1 weights on points -radius..radius
 * (src/common/fd_coeff2.cpp:49-57). */
YO_EXPORT void yo_center_fd_coefficients(double* coeff, int order, int radius) {
    int n = 2 * radius + 1;
    double* pts = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) pts[i] = (double)(i - radius);
    yo_fd_coeff(coeff, 0.0, order, pts, n);
    free(pts);
}

/* iso3dfd coefficient table c[0..radius]: c[0] = centre*3/2500, c[r] = w[r]/2500
 * (src/stencils/Iso3dfdStencil.cpp:68-90). */
YO_EXPORT void yo_iso3dfd_coeffs(double* c, int radius) {
    double w[2 * 64 + 1];
    yo_center_fd_coefficients(w, 2, radius);
    const double d2 = 50.0 * 50.0;
    for (int r = 0; r <= radius; r++) {
        double v = w[radius + r];
        if (r == 0) v *= 3.0;
        c[r] = v / d2;
    }
}

#define IDX(x, y, z) ((((int64_t)(x) + H) * sy_ + ((int64_t)(y) + H)) * sz_ + ((int64_t)(z) + H))

/* ------------------------------------------------------------------ typed bodies */
#define REAL float
#define SUF f32
#include "stencil_oracle_body.inc"
#undef REAL
#undef SUF
#define REAL double
#define SUF f64
#include "stencil_oracle_body.inc"
#undef REAL
#undef SUF
