// pt_bvh.cpp -- host-side BVH builder (see pt_bvh.h).  A binary tree first -- binned SAH near the root, object-median
// splits wherever SAH could make the tree deeper than the traversal stack --, then collapsed into nodes of up to four
// children wherever the stack budget allows.
#include "pt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace ptbvh {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
    void grow(const Box& b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    double half_area() const {
        double e[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
        if (e[0] < 0 || e[1] < 0 || e[2] < 0) return 0.0;
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
};

float down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f; }
float up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f; }

struct Prim {
    Box box;
    float cen[3];
    uint32_t obj;
};

// primitives a leaf is filled to (<= kMaxLeaf, what the traversal unrolls for); measurement knob
#ifndef PT_BVH_LEAF_TARGET
#define PT_BVH_LEAF_TARGET 4
#endif
constexpr uint32_t kLeafTarget = PT_BVH_LEAF_TARGET;
static_assert(kLeafTarget >= 1 && kLeafTarget <= kMaxLeaf, "leaf size");
// levels an object-median subtree of m primitives needs below its root
uint32_t median_levels(uint64_t m) {
    uint32_t l = 0;
    while (m > kLeafTarget) { m = (m + 1) / 2; ++l; }
    return l;
}

constexpr uint32_t kMaxDepth = kStackDepth - 2;   // deepest leaf the traversal stack (sentinel + one push per level) can take
constexpr int kBins = 16;

// node of the binary tree (temporary): child = leaf code, or index into Builder::bin
struct BinNode {
    Box box[2];
    uint32_t child[2];
    uint32_t height;          // levels of internal nodes below and including this one on its deepest path (a node of two leaves: 1)
};

struct Builder {
    std::vector<Prim> prims;
    std::vector<BinNode> bin;
    const float4* shape;
    const uint32_t* tag;
    Built out;

    uint32_t make_leaf(uint32_t first, uint32_t count, uint32_t depth) {
        while (out.leaf_ids.size() % 4u != 0u) {                 // leaves start at multiples of 4 slots
            out.leaf_ids.push_back(kDone);
            out.leaf_lead.push_back(make_float4(0, 0, 0, 0));
            for (int k = 0; k < 3; ++k) out.leaf_rec.push_back(make_float4(0, 0, 0, 0));
        }
        const uint32_t slot = (uint32_t)out.leaf_ids.size();
        // object order inside a leaf (not needed for correctness, keeps the tests in scan order)
        std::sort(prims.begin() + first, prims.begin() + first + count, [](const Prim& a, const Prim& b) { return a.obj < b.obj; });
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t o = prims[first + i].obj;
            const bool tri = tag[o] != 0;
            out.leaf_ids.push_back(o | (tri ? kTriangleBit : 0u));
            float4 r0 = shape[3 * (size_t)o], r1 = shape[3 * (size_t)o + 1], r2 = shape[3 * (size_t)o + 2];
            if (!tri) { r0.w = r0.w * r0.w; r1 = make_float4(0, 0, 0, 0); r2 = r1; }   // (c, r^2): the scan record of a sphere
            else { float4 t[3]; triangle_scan_record(r0, r1, r2, t); r0 = t[0]; r1 = t[1]; r2 = t[2]; }
            out.leaf_rec.push_back(r0); out.leaf_rec.push_back(r1); out.leaf_rec.push_back(r2);
            out.leaf_lead.push_back(r0);
            out.leaf_prims++;
        }
        out.depth = std::max(out.depth, depth);
        return kLeafBit | ((count - 1u) << 28) | slot;
    }

    // returns the child code of the subtree over prims[first, first+count)
    uint32_t build(uint32_t first, uint32_t count, uint32_t depth, Box* box_out) {
        Box box; box.reset();
        Box cb; cb.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            box.grow(prims[i].box);
            for (int k = 0; k < 3; ++k) { cb.lo[k] = std::min(cb.lo[k], prims[i].cen[k]); cb.hi[k] = std::max(cb.hi[k], prims[i].cen[k]); }
        }
        *box_out = box;
        if (count <= kLeafTarget) return make_leaf(first, count, depth);

        uint32_t mid = 0;
        bool split = false;
        if (depth + 1u + median_levels(count - 1u) <= kMaxDepth) {   // SAH may be arbitrarily unbalanced: only while that is safe
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const double lo = cb.lo[ax], ext = (double)cb.hi[ax] - lo;
                if (!(ext > 0.0)) continue;
                Box bb[kBins]; uint32_t bn[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bn[b] = 0; }
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = (int)(((double)prims[i].cen[ax] - lo) / ext * kBins);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(prims[i].box); bn[b]++;
                }
                double right_area[kBins]; uint32_t right_n[kBins];
                Box acc; acc.reset(); uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); n += bn[b]; right_area[b] = acc.half_area(); right_n[b] = n; }
                acc.reset(); n = 0;
                for (int b = 0; b + 1 < kBins; ++b) {
                    acc.grow(bb[b]); n += bn[b];
                    if (n == 0 || right_n[b + 1] == 0) continue;
                    const double cost = acc.half_area() * n + right_area[b + 1] * right_n[b + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const double lo = cb.lo[best_axis], ext = (double)cb.hi[best_axis] - lo;
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& p) {
                    int b = (int)(((double)p.cen[best_axis] - lo) / ext * kBins);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - prims.begin());
                split = mid > first && mid < first + count;
            }
        }
        if (!split) {   // object median along the widest centroid axis (or any axis when all centroids coincide)
            int ax = 0;
            double e = -1.0;
            for (int k = 0; k < 3; ++k) { const double x = (double)cb.hi[k] - cb.lo[k]; if (x > e) { e = x; ax = k; } }
            mid = first + (count + 1u) / 2u;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [ax](const Prim& a, const Prim& b) { return a.cen[ax] < b.cen[ax] || (a.cen[ax] == b.cen[ax] && a.obj < b.obj); });
        }
        const uint32_t node = (uint32_t)bin.size();
        bin.emplace_back();
        Box b0, b1;
        const uint32_t c0 = build(first, mid - first, depth + 1, &b0);
        const uint32_t c1 = build(mid, first + count - mid, depth + 1, &b1);
        BinNode& bn = bin[node];
        bn.box[0] = b0; bn.box[1] = b1;
        bn.child[0] = c0; bn.child[1] = c1;
        bn.height = 1u + std::max(height_of(c0), height_of(c1));
        return node;
    }
    uint32_t height_of(uint32_t code) const { return (code & kLeafBit) ? 0u : bin[code].height; }

    // Binary subtree `b` -> wide node; returns its index.  budget = stack entries the traversal may use below this
    // node's parent (need of the subtree <= budget).  A binary subtree of height h needs h entries as it is (one push per
    // level); pulling a grandchild pair up into the node costs one more entry for EVERY path through the node, so it is
    // done (largest box first) only while every child's binary height still fits what is left.
    uint32_t collapse(uint32_t b, uint32_t budget, uint32_t* need_out) {
        struct Item { Box box; uint32_t code; };
        Item it[kWidth];
        uint32_t c = 2;
        it[0] = {bin[b].box[0], bin[b].child[0]};
        it[1] = {bin[b].box[1], bin[b].child[1]};
        while (c < kWidth) {
            int best = -1;
            double best_area = -1.0;
            for (uint32_t k = 0; k < c; ++k) {
                if (it[k].code & kLeafBit) continue;
                // after opening child k the node has c + 1 children: every child subtree must fit budget - c
                bool ok = true;
                for (uint32_t j = 0; j < c && ok; ++j)
                    if (j != k) ok = height_of(it[j].code) + c <= budget;
                const BinNode& g = bin[it[k].code];
                ok = ok && height_of(g.child[0]) + c <= budget && height_of(g.child[1]) + c <= budget;
                if (!ok) continue;
                const double a = it[k].box.half_area();
                if (a > best_area) { best_area = a; best = (int)k; }
            }
            if (best < 0) break;
            const BinNode& g = bin[it[best].code];
            it[best] = {g.box[0], g.child[0]};
            it[c++] = {g.box[1], g.child[1]};
        }
        const uint32_t w = (uint32_t)out.wide.size();
        out.wide.emplace_back();
        uint32_t need_below = 0;
        for (uint32_t k = 0; k < kWidth; ++k) {
            uint32_t code = kDone;
            Box bx; bx.reset();
            if (k < c) {
                bx = it[k].box;
                code = it[k].code;
                if (!(code & kLeafBit)) {
                    uint32_t nd = 0;
                    code = collapse(code, budget - (c - 1u), &nd);
                    need_below = std::max(need_below, nd);
                }
            }
            WideNode& wn = out.wide[w];
            for (int a = 0; a < 3; ++a) { wn.lo[k][a] = bx.lo[a]; wn.hi[k][a] = bx.hi[a]; }
            wn.code[k] = code;
        }
        out.wide[w].n = c;
        *need_out = (c - 1u) + need_below;
        return w;
    }
};

// Child boxes -> 16-bit grid coordinates over the bounds of all child boxes (see pt_bvh.h).
void quantise(Built& t) {
    const size_t n_nodes = t.wide.size();
    t.qnodes.assign(4 * n_nodes, make_uint4(0, 0, 0, 0));
    if (n_nodes == 0) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const WideNode& w : t.wide)
        for (uint32_t c = 0; c < w.n; ++c)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)w.lo[c][a]); hi[a] = std::max(hi[a], (double)w.hi[c][a]); }
    for (int a = 0; a < 3; ++a) {
        t.grid_min[a] = down(lo[a]);
        const double ext = std::max(hi[a] - (double)t.grid_min[a], 1e-30);
        t.grid_cell[a] = up(ext / 65535.0 * (1.0 + 1e-6));          // 65535 cells reach past the upper bound
    }
    auto decode = [&](int a, uint32_t q) { return std::fmaf((float)q, t.grid_cell[a], t.grid_min[a]); };
    auto q_lo = [&](int a, float v) {
        long q = (long)std::floor(((double)v - t.grid_min[a]) / t.grid_cell[a]);
        q = std::min(std::max(q, 0L), 65535L);
        while (q > 0 && decode(a, (uint32_t)q) > v) --q;
        return (uint32_t)q;
    };
    auto q_hi = [&](int a, float v) {
        long q = (long)std::ceil(((double)v - t.grid_min[a]) / t.grid_cell[a]);
        q = std::min(std::max(q, 0L), 65535L);
        while (q < 65535 && decode(a, (uint32_t)q) < v) ++q;
        return (uint32_t)q;
    };
    for (size_t k = 0; k < n_nodes; ++k) {
        const WideNode& w = t.wide[k];
        uint32_t v[kWidth][3];
        for (uint32_t c = 0; c < kWidth; ++c) {
            if (c >= w.n) { v[c][0] = v[c][1] = v[c][2] = 0u; continue; }      // unused slot: its code (kDone) keeps the traversal out
            const uint32_t lx = q_lo(0, w.lo[c][0]), ly = q_lo(1, w.lo[c][1]), lz = q_lo(2, w.lo[c][2]);
            const uint32_t hx = q_hi(0, w.hi[c][0]), hy = q_hi(1, w.hi[c][1]), hz = q_hi(2, w.hi[c][2]);
            v[c][0] = lx | (ly << 16); v[c][1] = lz | (hx << 16); v[c][2] = hy | (hz << 16);
        }
        t.qnodes[4 * k] = make_uint4(v[0][0], v[0][1], v[0][2], v[1][0]);
        t.qnodes[4 * k + 1] = make_uint4(v[1][1], v[1][2], v[2][0], v[2][1]);
        t.qnodes[4 * k + 2] = make_uint4(v[2][2], v[3][0], v[3][1], v[3][2]);
        t.qnodes[4 * k + 3] = make_uint4(w.code[0], w.code[1], w.code[2], w.code[3]);
    }
}

}  // namespace

Built build(const float4* shape, const uint32_t* shape_tag, uint32_t n) {
    Builder b;
    b.shape = shape; b.tag = shape_tag;
    b.prims.resize(n);
    double amax[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; ++i) {
        Prim& p = b.prims[i];
        p.obj = i;
        const float4 r0 = shape[3 * (size_t)i], r1 = shape[3 * (size_t)i + 1], r2 = shape[3 * (size_t)i + 2];
        if (shape_tag[i] == 0) {
            // the scan tests against r2 = fl(r*r); bound the sphere of radius sqrt(r2), rounded outward
            const float r2f = r0.w * r0.w;
            const double r = std::sqrt((double)r2f) * (1.0 + 1e-7);
            const double c[3] = {r0.x, r0.y, r0.z};
            for (int k = 0; k < 3; ++k) { p.box.lo[k] = down(c[k] - r); p.box.hi[k] = up(c[k] + r); }
        } else {
            const double v0[3] = {r0.x, r0.y, r0.z}, e1[3] = {r1.x, r1.y, r1.z}, e2[3] = {r2.x, r2.y, r2.z};
            for (int k = 0; k < 3; ++k) {
                const double a = v0[k], bq = v0[k] + e1[k], c = v0[k] + e2[k];
                p.box.lo[k] = down(std::min(a, std::min(bq, c)));
                p.box.hi[k] = up(std::max(a, std::max(bq, c)));
            }
        }
        bool finite = true;
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(p.box.lo[k]) && std::isfinite(p.box.hi[k]);
        if (!finite) b.out.non_finite++;
        for (int k = 0; k < 3; ++k) {
            // a non-finite box would poison every ancestor: make it cover everything (the caller refuses the scene anyway)
            if (!finite) {
                p.box.lo[k] = -std::numeric_limits<float>::max(); p.box.hi[k] = std::numeric_limits<float>::max();
                p.cen[k] = 0.f;
            } else {
                p.cen[k] = (float)(0.5 * ((double)p.box.lo[k] + p.box.hi[k]));
                amax[k] = std::max(amax[k], std::max(std::fabs((double)p.box.lo[k]), std::fabs((double)p.box.hi[k])));
            }
        }
    }
    b.bin.reserve(n);
    b.out.leaf_ids.reserve(n);
    b.out.leaf_rec.reserve(3 * (size_t)n);
    if (n != 0) {
        Box root;
        uint32_t r = b.build(0, n, 0, &root);
        if (!(r & kLeafBit)) {
            // the binary tree is at most kMaxDepth deep: as it stands it fits the budget, and the collapse never breaks that
            uint32_t need = 0;
            r = b.collapse(r, kStackDepth - 2u, &need);
            b.out.stack_need = 1u + need;
        }
        b.out.root = r;
    }
    b.out.scene_abs = up(amax[0] + amax[1] + amax[2]);
    while (b.out.leaf_ids.size() % 4u != 0u) {                   // the last leaf's 16-byte id load stays in bounds
        b.out.leaf_ids.push_back(kDone);
        b.out.leaf_lead.push_back(make_float4(0, 0, 0, 0));
        for (int k = 0; k < 3; ++k) b.out.leaf_rec.push_back(make_float4(0, 0, 0, 0));
    }
    quantise(b.out);
    return std::move(b.out);
}

}  // namespace ptbvh
