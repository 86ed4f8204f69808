// CPU driver of csrc/ykh_boxes.hpp (tests/test_part_boxes_cpu.py): regions given as predicates over a small grid, the two reductions
// answered by brute force; prints, per region, whether a list was found and whether it is exact (full, disjoint, covering).
#include <algorithm>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
#include "ykh_boxes.hpp"

struct B { long long lo[3], hi[3]; };
typedef std::function<bool(long long, long long, long long)> Pred;

static void run(const char* name, const long long n[3], Pred in, bool first) {
    unsigned long long queries = 0, profiles = 0;
    auto query = [&](const B& q, B& bb) -> unsigned long long {
        queries++;
        unsigned long long c = 0;
        long long lo[3] = {1LL << 40, 1LL << 40, 1LL << 40}, hi[3] = {-1, -1, -1};
        for (long long x = q.lo[0]; x < q.hi[0]; x++)
            for (long long y = q.lo[1]; y < q.hi[1]; y++)
                for (long long z = q.lo[2]; z < q.hi[2]; z++)
                    if (in(x, y, z)) {
                        c++;
                        const long long p[3] = {x, y, z};
                        for (int d = 0; d < 3; d++) { if (p[d] < lo[d]) lo[d] = p[d]; if (p[d] > hi[d]) hi[d] = p[d]; }
                    }
        bb = q;
        if (c) for (int d = 0; d < 3; d++) { bb.lo[d] = lo[d]; bb.hi[d] = hi[d] + 1; }
        return c;
    };
    auto profile = [&](const B& bb, std::vector<unsigned>& h) {
        profiles++;
        const long long nx = bb.hi[0] - bb.lo[0], ny = bb.hi[1] - bb.lo[1];
        for (long long x = bb.lo[0]; x < bb.hi[0]; x++)
            for (long long y = bb.lo[1]; y < bb.hi[1]; y++)
                for (long long z = bb.lo[2]; z < bb.hi[2]; z++)
                    if (in(x, y, z)) { h[x - bb.lo[0]]++; h[nx + y - bb.lo[1]]++; h[nx + ny + z - bb.lo[2]]++; }
    };
    B whole = {{0, 0, 0}, {n[0], n[1], n[2]}}, bb0;
    const unsigned long long total = query(whole, bb0);
    std::vector<B> out;
    const bool ok = total > 0 && ykh::decompose_full_boxes<B>(bb0, total, query, profile, out);
    // exactness of whatever came out
    std::vector<int> cover((size_t)(n[0] * n[1] * n[2]), 0);
    bool inside = true;
    for (const B& b : out)
        for (long long x = b.lo[0]; x < b.hi[0]; x++)
            for (long long y = b.lo[1]; y < b.hi[1]; y++)
                for (long long z = b.lo[2]; z < b.hi[2]; z++) { cover[(size_t)((x * n[1] + y) * n[2] + z)]++; inside = inside && in(x, y, z); }
    bool disjoint = true, all = true;
    for (long long x = 0; x < n[0]; x++)
        for (long long y = 0; y < n[1]; y++)
            for (long long z = 0; z < n[2]; z++) {
                const int c = cover[(size_t)((x * n[1] + y) * n[2] + z)];
                if (c > 1) disjoint = false;
                if (in(x, y, z) && c == 0) all = false;
            }
    printf("%s{\"region\": \"%s\", \"found\": %d, \"boxes\": %zu, \"valid_points\": %llu, \"full\": %d, \"disjoint\": %d, \"covering\": %d, \"queries\": %llu, \"profiles\": %llu}",
           first ? "" : ",\n", name, (int)ok, out.size(), total, (int)inside, (int)disjoint, (int)all, queries, profiles);
}

int main() {
    const long long n[3] = {40, 36, 48};
    printf("[");
    // an absorbing shell of width 5 (fsg_abc: FSGElasticStencil.cpp:395-397 with width 20)
    run("shell", n, [&](long long x, long long y, long long z) { return x < 5 || x >= n[0] - 5 || y < 5 || y >= n[1] - 5 || z < 5 || z >= n[2] - 5; }, true);
    // the complement of a box (test_boundary_3d's !sd0, TestStencils.cpp:853-865)
    run("box_complement", n, [&](long long x, long long y, long long z) { return !(x >= 5 && x <= n[0] - 4 && y >= 4 && y <= n[1] - 7 && z >= 6 && z <= n[2] - 5); }, false);
    // one plane minus its margins (a free surface inside the sponge layers)
    run("plane_with_margins", n, [&](long long x, long long y, long long z) { return z == n[2] - 3 && x >= 4 && x < n[0] - 4 && y >= 4 && y < n[1] - 4; }, false);
    // two slabs that overlap in a corner (an L)
    run("l_shape", n, [&](long long x, long long y, long long z) { return x < 10 || y < 8; }, false);
    // two separate boxes
    run("two_boxes", n, [&](long long x, long long y, long long z) { return (x < 7 && y < 9 && z < 11) || (x >= 20 && x < 30 && y >= 20 && z >= 30); }, false);
    // a region that fills its bounding box
    run("solid", n, [&](long long x, long long y, long long z) { return x >= 3 && x < 17 && z >= 2; }, false);
    // not a handful of slabs: a slanted half-space, a checkerboard, a staircase of 12 steps
    run("slanted", n, [&](long long x, long long y, long long z) { return x < y; }, false);
    run("checkerboard", n, [&](long long x, long long y, long long z) { return ((x + y + z) & 1) == 0; }, false);
    run("staircase", n, [&](long long x, long long y, long long z) { return y <= (x / 3) * 3; }, false);
    // ---- rings of 2-D solutions (ring_strips): a 40 x 36 grid, the region known through the same two reductions
    {
        const long long m[2] = {40, 36};
        auto ring = [&](const char* name, auto in) {
            B bb{{m[0], m[1], 0}, {0, 0, 1}}, hole{{m[0], m[1], 0}, {0, 0, 1}};
            unsigned long long nt = 0, nf = 0;
            for (long long x = 0; x < m[0]; x++)
                for (long long y = 0; y < m[1]; y++)
                    if (in(x, y)) { nt++; bb.lo[0] = std::min(bb.lo[0], x); bb.hi[0] = std::max(bb.hi[0], x + 1); bb.lo[1] = std::min(bb.lo[1], y); bb.hi[1] = std::max(bb.hi[1], y + 1); }
            for (long long x = bb.lo[0]; x < bb.hi[0]; x++)
                for (long long y = bb.lo[1]; y < bb.hi[1]; y++)
                    if (!in(x, y)) { nf++; hole.lo[0] = std::min(hole.lo[0], x); hole.hi[0] = std::max(hole.hi[0], x + 1); hole.lo[1] = std::min(hole.lo[1], y); hole.hi[1] = std::max(hole.hi[1], y + 1); }
            std::vector<B> out;
            const bool ok = ykh::ring_strips(bb, nt, hole, nf, out);
            std::vector<int> cover((size_t)(m[0] * m[1]), 0);
            bool inside = true, disjoint = true, all = true;
            for (const B& b : out)
                for (long long x = b.lo[0]; x < b.hi[0]; x++)
                    for (long long y = b.lo[1]; y < b.hi[1]; y++) { cover[(size_t)(x * m[1] + y)]++; inside = inside && in(x, y); }
            for (long long x = 0; x < m[0]; x++)
                for (long long y = 0; y < m[1]; y++) {
                    if (cover[(size_t)(x * m[1] + y)] > 1) disjoint = false;
                    if (in(x, y) && cover[(size_t)(x * m[1] + y)] == 0) all = false;
                }
            printf(",\n{\"region\": \"%s\", \"found\": %d, \"boxes\": %zu, \"valid_points\": %llu, \"full\": %d, \"disjoint\": %d, \"covering\": %d, \"queries\": 2, \"profiles\": 0}",
                   name, (int)ok, out.size(), nt, (int)inside, (int)disjoint, (int)(all || !ok));
        };
        // the boundary ring around an interior (swe2d / wave2d: the complement of `x > first + 1 && x < last - 1 && ...`)
        ring("ring2d", [&](long long x, long long y) { return !(x >= 2 && x < m[0] - 2 && y >= 3 && y < m[1] - 1); });
        // the interior touches one side of the grid: three strips
        ring("ring2d_open_side", [&](long long x, long long y) { return !(x >= 0 && x < m[0] - 4 && y >= 3 && y < m[1] - 3); });
        // not rings: two separate holes; an L-shaped hole; a solid box (no hole at all)
        ring("two_holes2d", [&](long long x, long long y) { return !((x >= 3 && x < 8 && y >= 3 && y < 8) || (x >= 20 && x < 25 && y >= 20 && y < 25)); });
        ring("l_hole2d", [&](long long x, long long y) { return !((x >= 5 && x < 30 && y >= 5 && y < 10) || (x >= 5 && x < 10 && y >= 5 && y < 30)); });
        ring("solid2d", [&](long long x, long long y) { return x >= 4 && y >= 4; });
    }
    printf("]\n");
    return 0;
}
