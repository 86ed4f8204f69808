// ykh_boxes.hpp -- full boxes of a region known only through two reductions (host logic of Solution::find_part_boxes; pure C++, no HIP:
// tests/cpp/boxes_probe.cpp drives it with a mask on the CPU).
//
// The reference turns a sub-domain condition into a list of FULL bounding boxes -- non-overlapping rectangles holding valid points
// only -- by scanning every point on the host (StencilPartBase::find_bounding_boxes, src/kernel/lib/setup.cpp:1235-1500: slices per
// thread, rectangles grown point by point, merged afterwards).  Here the condition is device code and the grid may hold 2^30
// points, so the region is never enumerated.  Two questions are asked of it instead:
//   query(q, bb)      -> number of valid points inside box q, and their bounding box bb (count == volume(bb): bb is full);
//   profile(bb, hist) -> number of valid points per index along x, then y, then z of box bb (hist has nx + ny + nz entries).
// A profile is constant between the planes that bound the region, so a box that is not full is cut at the breakpoints of the dim
// that has the fewest of them and the pieces are examined again.  Regions made of a few slabs (a shell of fixed width: 6 boxes)
// resolve in three levels; anything that does not -- a slanted or scattered condition -- is given up (returns false: the caller
// keeps the point kernel's per-point predicate).  Exact by construction and checked: every emitted box is full and the volumes
// add up to the number of valid points.
#pragma once
#include <cstddef>
#include <vector>

namespace ykh {

struct BoxListLimits { int max_boxes = 64, max_depth = 8, max_cuts = 8; };

// BoxT: anything with lo[3] / hi[3] (half-open).  Query: unsigned long long(const BoxT& q, BoxT& bb).
// Profile: void(const BoxT& bb, std::vector<unsigned>& hist).
template <class BoxT, class Query, class Profile>
bool decompose_full_boxes(const BoxT& bb0, unsigned long long total, Query&& query, Profile&& profile, std::vector<BoxT>& out,
                          const BoxListLimits lim = BoxListLimits()) {
    auto volume = [](const BoxT& b) {
        unsigned long long v = 1;
        for (int d = 0; d < 3; d++) v *= (unsigned long long)(b.hi[d] > b.lo[d] ? b.hi[d] - b.lo[d] : 0);
        return v;
    };
    struct Item { BoxT b; int depth; bool known; unsigned long long count; };
    std::vector<Item> todo;
    todo.push_back(Item{bb0, 0, true, total});
    std::vector<unsigned> hist;
    unsigned long long covered = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        BoxT bb = it.b;
        const unsigned long long c = it.known ? it.count : query(it.b, bb);
        if (c == 0) continue;
        if (c == volume(bb)) {
            out.push_back(bb);
            covered += c;
            if ((int)out.size() > lim.max_boxes) return false;
            continue;
        }
        if (it.depth >= lim.max_depth) return false;
        const long long n[3] = {(long long)(bb.hi[0] - bb.lo[0]), (long long)(bb.hi[1] - bb.lo[1]), (long long)(bb.hi[2] - bb.lo[2])};
        hist.assign((size_t)(n[0] + n[1] + n[2]), 0u);
        profile(bb, hist);
        // breakpoints per dim; cut along the dim with the fewest pieces (> 1)
        int best = -1;
        std::vector<long long> cuts[3];
        for (int d = 0; d < 3; d++) {
            const unsigned* h = hist.data() + (d == 0 ? 0 : (d == 1 ? n[0] : n[0] + n[1]));
            for (long long i = 1; i < n[d]; i++)
                if (h[i] != h[i - 1]) cuts[d].push_back(i);
            if (!cuts[d].empty() && (int)cuts[d].size() < lim.max_cuts && (best < 0 || cuts[d].size() < cuts[best].size())) best = d;
        }
        if (best < 0) return false;
        long long from = 0;
        cuts[best].push_back(n[best]);
        for (long long to : cuts[best]) {
            BoxT piece = bb;
            piece.lo[best] = bb.lo[best] + from;
            piece.hi[best] = bb.lo[best] + to;
            todo.push_back(Item{piece, it.depth + 1, false, 0});
            from = to;
        }
    }
    return covered == total && !out.empty();
}

// A ring: a box minus a box (2-D solutions; Solution::prepare).  The complement of an interior condition inside a grid is the boundary
// ring around the interior -- not a solid box, yet two reductions describe it exactly: count + bounding box `bb` of the points where the
// condition holds, and count + bounding box `hole` of the points of bb where it does NOT.  If the hole is full (n_false == volume(hole))
// and nothing else is missing (n_true + n_false == volume(bb)), the region is bb \ hole: at most four strips, emitted here.
// Returns false (no strips) when the region is not such a ring.  Dims 0 and 1; dim 2 of every strip is bb's.
template <class BoxT>
bool ring_strips(const BoxT& bb, unsigned long long n_true, const BoxT& hole, unsigned long long n_false, std::vector<BoxT>& out) {
    auto vol2 = [](const BoxT& b) {
        unsigned long long v = 1;
        for (int d = 0; d < 2; d++) v *= (unsigned long long)(b.hi[d] > b.lo[d] ? b.hi[d] - b.lo[d] : 0);
        return v;
    };
    if (n_false == 0 || n_false != vol2(hole) || n_true + n_false != vol2(bb)) return false;
    for (int d = 0; d < 2; d++)
        if (hole.lo[d] < bb.lo[d] || hole.hi[d] > bb.hi[d]) return false;
    auto add = [&](long long l0, long long h0, long long l1, long long h1) {
        BoxT b = bb;
        b.lo[0] = l0; b.hi[0] = h0; b.lo[1] = l1; b.hi[1] = h1;
        if (b.hi[0] > b.lo[0] && b.hi[1] > b.lo[1]) out.push_back(b);
    };
    add(bb.lo[0], hole.lo[0], bb.lo[1], bb.hi[1]);          // rows before the hole
    add(hole.hi[0], bb.hi[0], bb.lo[1], bb.hi[1]);          // rows after it
    add(hole.lo[0], hole.hi[0], bb.lo[1], hole.lo[1]);      // left of it
    add(hole.lo[0], hole.hi[0], hole.hi[1], bb.hi[1]);      // right of it
    return true;
}

}  // namespace ykh
