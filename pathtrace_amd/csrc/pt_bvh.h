// pt_bvh.h -- optional acceleration structure for World::hit_scene (the reference
// itself is linear-scan only, world.rs:281).  A 4-wide BVH over the objects' f32 bounding boxes (a binned-SAH
// binary tree, collapsed), built on the host at first use.  It only decides WHICH primitives a ray is tested against; the
// primitive tests are the ones of the linear scan, and the winner is chosen by the rule the scan
// implies (smallest t; among equal t the highest object index), so the answer does not depend on
// the traversal order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace ptbvh {

// Child code of a node: bit 31 clear = index of an internal node; bit 31 set = leaf with
// ((code >> 28) & 7) + 1 primitives starting at slot (code & 0x0FFFFFFF) of the leaf arrays.
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kDone = 0xFFFFFFFFu;      // stack sentinel / root of an empty scene (no leaf code: a leaf holds <= 4 primitives)
constexpr uint32_t kMaxLeaf = 4;             // primitives per leaf
constexpr uint32_t kWidth = 4;               // children per internal node
#ifndef PT_BVH_STACK
#define PT_BVH_STACK 24
#endif
constexpr uint32_t kStackDepth = PT_BVH_STACK;         // traversal stack entries per lane (LDS); the builder keeps the tree's stack need within it (<= 16 M objects)
constexpr uint32_t kTriangleBit = 0x80000000u;   // in leaf_ids: the primitive is a triangle

// Internal node as the builder and the checker see it: up to kWidth children, each with the f32 box of its subtree.
// Unused child slots carry the code kDone (the traversal never enters them).
struct WideNode {
    float lo[kWidth][3], hi[kWidth][3];
    uint32_t code[kWidth];
    uint32_t n;                       // children in use (2 .. kWidth), slots [0, n)
};

struct Built {
    std::vector<WideNode> wide;
    // What the device traverses: the same nodes with the child boxes on a 16-bit grid over the scene's bounds, 64 bytes
    // (one cache line, four 16-byte loads) per visit.  Round 2's binary nodes took two dependent 32-byte visits for what
    // one visit decides here; the traversal is bound by the latency of those dependent fetches.
    //   child c = three words  w0 = lo.x | lo.y << 16, w1 = lo.z | hi.x << 16, w2 = hi.y | hi.z << 16
    //   qnodes[4k]   = (c0.w0, c0.w1, c0.w2, c1.w0)    qnodes[4k+1] = (c1.w1, c1.w2, c2.w0, c2.w1)
    //   qnodes[4k+2] = (c2.w2, c3.w0, c3.w1, c3.w2)    qnodes[4k+3] = (code0, code1, code2, code3)
    // coordinate = grid_min[axis] + q * grid_cell[axis], evaluated as fmaf((float)q, cell, min); lower planes are
    // rounded down and upper planes up until that f32 expression encloses the f32 box, so a quantised box contains
    // the exact one (it only prunes less).
    std::vector<uint4> qnodes;
    float grid_min[3] = {0.f, 0.f, 0.f}, grid_cell[3] = {0.f, 0.f, 0.f};
    // Leaf slots.  Every leaf starts at a slot index that is a multiple of 4 (unused slots: id kDone), so the <= 4 ids of
    // a leaf are one aligned 16-byte load and its <= 4 lead records one 64-byte line.
    std::vector<float4> leaf_rec;     // 3 float4 per leaf slot: sphere (c, r^2), -, - ; triangle: triangle_scan_record (the scan records)
    std::vector<float4> leaf_lead;    // 1 float4 per leaf slot = leaf_rec[3 * slot]: all a sphere test reads (a triangle reads the other two from leaf_rec)
    std::vector<uint32_t> leaf_ids;   // object index of the leaf slot (| kTriangleBit)
    uint32_t leaf_prims = 0;          // slots that hold a primitive (= number of objects)
    uint32_t root = kDone;            // child code of the root
    uint32_t depth = 0;               // deepest leaf of the BINARY tree the nodes were collapsed from (root = 0)
    // Stack entries a traversal can need: 1 (sentinel) + the largest sum over a root-to-leaf path of (children - 1): a
    // visit pushes every hit child but the one it descends into.  The collapse keeps it <= kStackDepth (it merges a
    // binary node's grandchildren into the node only where the remaining budget still covers the subtrees below).
    uint32_t stack_need = 1;
    float scene_abs = 0.0f;           // sum over axes of the largest |coordinate| of any box: scale of the traversal padding
    // Objects with a NaN/inf coordinate or radius.  The linear scan's answer for such an object depends on the
    // scan order (a NaN t is "accepted" and then lets every later hit through, world.rs:281-287), which no
    // traversal order reproduces: callers refuse accel = 1 for such scenes.
    uint32_t non_finite = 0;
};

// Scan record of a triangle (what the primitive test reads; pt_kernels.hip triangle_test): the f32 specification of
// TriangleShape::hit (shape.rs:161-192) works on the triangle's plane and two barycentric gradients instead of
// re-deriving them per ray from the edges as Moeller-Trumbore does -- same real-number u, v, t:
//     n  = e1 x e2                 t = -(s.n) / (d.n),  s = o - v0      (a = e1.(d x e2) = -(d.n), t = f e2.(s x e1) = f s.n)
//     N1 = (e2 x n) / (n.n)        u = (s + t d) . N1                    (N1.e1 = 1, N1.e2 = 0, N1.n = 0)
//     N2 = (n x e1) / (n.n)        v = (s + t d) . N2                    (N2.e2 = 1, N2.e1 = 0, N2.n = 0)
// computed in f64 from the f32 edges and rounded to f32 (the oracle's float instantiation does the same, bit for bit).
// Packing, 3 float4, in the order the test reads them -- one aligned 16-byte read per stage (round 5; rounds 3-4 packed v0 first
// and the compiler read the normal as two 8-byte halves of two float4): (n.x, n.y, n.z, N1.x) (v0.x, v0.y, v0.z, N1.y)
// (N1.z, N2.x, N2.y, N2.z).
inline void triangle_scan_record(const float4& v0, const float4& e1f, const float4& e2f, float4 out[3]) {
    const double e1[3] = {e1f.x, e1f.y, e1f.z}, e2[3] = {e2f.x, e2f.y, e2f.z};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double a[3] = {e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0]};   // e2 x n
    const double b[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};   // n x e1
    out[0] = make_float4((float)n[0], (float)n[1], (float)n[2], (float)(a[0] / nn));
    out[1] = make_float4(v0.x, v0.y, v0.z, (float)(a[1] / nn));
    out[2] = make_float4((float)(a[2] / nn), (float)(b[0] / nn), (float)(b[1] / nn), (float)(b[2] / nn));
}

// shape: 3 float4 per object in the gather form of pt_device.h (sphere: (c, r), (1/r,..), -; triangle: v0, e1, e2);
// scan_w: for spheres the r^2 the scan record carries.  Throws nothing; n may be 0.
Built build(const float4* shape, const uint32_t* shape_tag, uint32_t n);

}  // namespace ptbvh
