// bvh.hip — LBVH construction on the device (gfx950).
//
// The reference has no acceleration structure: UScene::intersectionPoint / visibility scan every
// object and Mesh::localIntersection every face (uengine/uscene.cpp:21-66, uapp/mesh.cpp:108-189).
// This builds one world-space tree over all triangles and spheres:
//   world boxes (fp64) -> 63-bit Morton codes of the box centres -> radix sort -> Karras 2012
//   hierarchy -> level-synchronous bottom-up refit of the binary tree -> top-down collapse into wide nodes
//   (the inner child with the largest surface area is opened until a node has kNodeWidth children; against
//   merging every second level this cut the node visits per ray on a 2 M-triangle scene from 9.3 to 8.4)
//   -> child boxes quantised to six bytes inside each node (scene.h).
// Node boxes are fp32, rounded outward and padded by 2^-20 of the scene size, so the fp32 slab test
// in trace.h never rejects a primitive the reference's object-local fp64 test would accept
// (local and world coordinates differ only by the rounding of W and its inverse).
// The refit runs one launch per tree level: a node is produced only from children finished by an
// EARLIER launch, so no intra-launch cross-workgroup visibility is ever relied on.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>

#include "render.h"

namespace ub {

namespace {

constexpr int kB = 256;

struct Box6 { double lo[3], hi[3]; };
struct FBox { float lo[3], hi[3]; };
// node of the binary tree the builder works on: boxes and links of both children
struct BinNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
};

__device__ __forceinline__ unsigned long long orderedBits(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double fromOrdered(unsigned long long b) {
    b = (b & 0x8000000000000000ULL) ? (b & 0x7fffffffffffffffULL) : ~b;
    return __longlong_as_double((long long)b);
}

// world-space bounds of every primitive; scene bounds by ordered-integer atomics
__global__ void k_prim_bounds(const DObject* objects, const Prim* prims, uint32_t n, Box6* boxes,
                              unsigned long long* scene /*[6]: lo xyz as min, hi xyz as max*/) {
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const Prim pr = prims[i];
    const DObject& ob = objects[pr.object];
    Box6 b;
    if (pr.face == kSphereFace) {
        // W maps the unit sphere to an ellipsoid: centre W[3], half extent on axis k = |row k of the 3x3|
        const double* c0 = &ob.W.c[0].x; const double* c1 = &ob.W.c[1].x; const double* c2 = &ob.W.c[2].x;
        const double* c3 = &ob.W.c[3].x;
        for (int k = 0; k < 3; k++) {
            double h = sqrt(c0[k] * c0[k] + c1[k] * c1[k] + c2[k] * c2[k]);
            h = h * (1.0 + 1e-12) + 1e-300;
            b.lo[k] = c3[k] - h;
            b.hi[k] = c3[k] + h;
        }
    } else {
        for (int k = 0; k < 3; k++) { b.lo[k] = __builtin_inf(); b.hi[k] = -__builtin_inf(); }
        for (int v = 0; v < 3; v++) {
            d3 p = mul_xyz(ob.W, mk3((double)pr.p[3 * v], (double)pr.p[3 * v + 1], (double)pr.p[3 * v + 2]), 1.0);
            b.lo[0] = fmin(b.lo[0], p.x); b.hi[0] = fmax(b.hi[0], p.x);
            b.lo[1] = fmin(b.lo[1], p.y); b.hi[1] = fmax(b.hi[1], p.y);
            b.lo[2] = fmin(b.lo[2], p.z); b.hi[2] = fmax(b.hi[2], p.z);
        }
    }
    boxes[i] = b;
    for (int k = 0; k < 3; k++) {
        atomicMin(&scene[k], orderedBits(b.lo[k]));
        atomicMax(&scene[3 + k], orderedBits(b.hi[k]));
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x) {
    x &= 0x1fffffULL;
    x = (x | (x << 32)) & 0x1f00000000ffffULL;
    x = (x | (x << 16)) & 0x1f0000ff0000ffULL;
    x = (x | (x << 8)) & 0x100f00f00f00f00fULL;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x1249249249249249ULL;
    return x;
}

__global__ void k_morton(const Box6* boxes, uint32_t n, const unsigned long long* scene, unsigned long long* keys,
                         uint32_t* idx) {
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    unsigned long long code = 0;
    for (int k = 0; k < 3; k++) {
        double lo = fromOrdered(scene[k]), hi = fromOrdered(scene[3 + k]);
        double c = 0.5 * (boxes[i].lo[k] + boxes[i].hi[k]);
        double ext = hi - lo;
        double t = ext > 0 ? (c - lo) / ext : 0.0;
        t = fmin(fmax(t, 0.0), 1.0);
        unsigned long long q = (unsigned long long)(t * 2097151.0);
        code |= spread21(q) << (2 - k);
    }
    keys[i] = code;
    idx[i] = i;
}

// boxes to fp32, outward, padded; primitives into tree order
__global__ void k_gather(const Prim* prims_in, const Box6* boxes, const uint32_t* idx, uint32_t n,
                         const unsigned long long* scene, Prim* prims_out, FBox* leaf_boxes) {
    uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    uint32_t src = idx[i];
    prims_out[i] = prims_in[src];
    double mag = 0;
    for (int k = 0; k < 3; k++) {
        double lo = fromOrdered(scene[k]), hi = fromOrdered(scene[3 + k]);
        mag = fmax(mag, fmax(fabs(lo), fabs(hi)));
    }
    double pad = mag * (1.0 / 1048576.0) + 1e-30;
    FBox f;
    for (int k = 0; k < 3; k++) {
        f.lo[k] = __double2float_rd(boxes[src].lo[k] - pad);
        f.hi[k] = __double2float_ru(boxes[src].hi[k] + pad);
    }
    leaf_boxes[i] = f;
}

// Karras 2012: common-prefix length of sorted keys i and j, ties broken by index
__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void k_hierarchy(const unsigned long long* keys, int n, int32_t* child0, int32_t* child1) {
    int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    child0[i] = (lo == gamma) ? ~gamma : gamma;
    child1[i] = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
}


// ---- locally-ordered clustering on the Morton order (Meister and Bittner 2018) ------------------------------------
// The Karras hierarchy splits the sorted run wherever the codes' highest differing bit says so, whatever the boxes look
// like. Here the sorted primitives start as one cluster each, and in every round a cluster looks kPlocRadius places to
// the left and right along the Morton order for the partner whose union with it has the smallest surface area; two
// clusters that choose each other are merged into a node. Rounds repeat until one cluster is left. Candidates are
// ordered by (area, distance along the order, parity of the lower position, lower position): a total order on pairs,
// so the smallest pair in reach always chooses itself (every round merges something), and runs of equal boxes pair up
// even-odd instead of forming chains in which only the first pair is mutual.
constexpr int kPlocRadius = 8;      // default reach of the search; UBDPT_PLOC_RADIUS overrides it for measurements
constexpr int kPlocMaxRadius = 64;

__device__ __forceinline__ float unionArea(const FBox& a, const FBox& b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kB) void k_ploc_nearest(const FBox* boxes, uint32_t m, uint32_t* nn, int radius) {
    __shared__ FBox tile[kB + 2 * kPlocMaxRadius];
    const int base = (int)(blockIdx.x * kB) - radius;
    for (int t = threadIdx.x; t < kB + 2 * radius; t += kB) {
        const int g = base + t;
        if (g >= 0 && g < (int)m) tile[t] = boxes[g];
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m) return;
    const FBox me = tile[threadIdx.x + radius];
    float best_area = __builtin_inff();
    uint32_t best_dist = 0xffffffffu, best_par = 2, best_lo = 0xffffffffu, best = i;
    for (int d = -radius; d <= radius; d++) {
        if (d == 0) continue;
        const long long j = (long long)i + d;
        if (j < 0 || j >= (long long)m) continue;
        const float area = unionArea(me, tile[(int)threadIdx.x + radius + d]);
        const uint32_t dist = (uint32_t)(d < 0 ? -d : d), lo = (uint32_t)(d < 0 ? j : i), par = lo & 1u;
        const bool less = area < best_area ||
                          (area == best_area && (dist < best_dist || (dist == best_dist && (par < best_par || (par == best_par && lo < best_lo)))));
        if (less) { best_area = area; best_dist = dist; best_par = par; best_lo = lo; best = (uint32_t)j; }
    }
    nn[i] = best;
}
// flag word of position i: low half 1 when a cluster stays at i after this round (everything but the upper partner of a
// merge), high half 1 when i is the lower partner of a merge and so makes a node
__global__ void k_ploc_flags(const uint32_t* nn, uint32_t m, unsigned long long* flags) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    flags[i] = (mutual && j < i) ? 0ULL : (1ULL | ((mutual && i < j) ? (1ULL << 32) : 0ULL));
}
// writes the next round's clusters (compacted, order kept) and the nodes of this round's merges; the node of the
// merge with rank k among this round's merges gets index id_top - 1 - k, so the last merge of all becomes node 0
__global__ void k_ploc_merge(const uint32_t* nn, const unsigned long long* flags, const unsigned long long* scan, uint32_t m,
                             const FBox* boxes, const int32_t* ids, FBox* boxes_out, int32_t* ids_out, uint32_t id_top,
                             int32_t* child0, int32_t* child1) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ULL)) return;
    const uint32_t pos = (uint32_t)(scan[i] & 0xffffffffULL);
    if (f >> 32) {
        const uint32_t j = nn[i];
        const uint32_t node = id_top - 1u - (uint32_t)(scan[i] >> 32);
        child0[node] = ids[i];
        child1[node] = ids[j];
        const FBox a = boxes[i], b = boxes[j];
        FBox u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
        boxes_out[pos] = u;
        ids_out[pos] = (int32_t)node;
    } else {
        boxes_out[pos] = boxes[i];
        ids_out[pos] = ids[i];
    }
}
__global__ void k_ploc_init(uint32_t n, int32_t* ids) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < n) ids[i] = ~(int32_t)i;
}

// One refit level: node i is produced when both children were finished by an earlier launch.
// ready[i] = number of the launch that produced node i (0 = not yet).
__global__ void k_refit_level(int n_internal, const int32_t* child0, const int32_t* child1, const FBox* leaf_boxes,
                              FBox* node_boxes, uint32_t* ready, uint32_t level, BinNode* nodes, uint32_t* progress) {
    int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_internal) return;
    if (ready[i] != 0) return;
    int32_t c0 = child0[i], c1 = child1[i];
    if (c0 >= 0) { uint32_t r = ready[c0]; if (r == 0 || r >= level) return; }
    if (c1 >= 0) { uint32_t r = ready[c1]; if (r == 0 || r >= level) return; }
    FBox b0 = c0 < 0 ? leaf_boxes[~c0] : node_boxes[c0];
    FBox b1 = c1 < 0 ? leaf_boxes[~c1] : node_boxes[c1];
    BinNode nd;
    FBox u;
    for (int k = 0; k < 3; k++) {
        nd.lo0[k] = b0.lo[k]; nd.hi0[k] = b0.hi[k];
        nd.lo1[k] = b1.lo[k]; nd.hi1[k] = b1.hi[k];
        u.lo[k] = fminf(b0.lo[k], b1.lo[k]);
        u.hi[k] = fmaxf(b0.hi[k], b1.hi[k]);
    }
    nd.child0 = c0; nd.child1 = c1;
    nodes[i] = nd;
    node_boxes[i] = u;
    ready[i] = level;
    atomicAdd(progress, 1u);
}

// Biased exponent of the quantisation step of a node along one axis: the smallest power of two with 254 steps
// covering [o, top] (0: the node is flat there and the step is 0).
__device__ uint32_t stepExponent(float o, float top) {
    const double extent = (double)top - (double)o;  // exact
    if (!(extent > 0.0)) return 0;
    int ex;
    const double m = frexp(extent / 254.0, &ex);  // extent / 254 = m * 2^ex, m in [0.5, 1)
    const int biased = (m == 0.5 ? ex - 1 : ex) + 127;
    return (uint32_t)(biased < 1 ? 1 : (biased > 254 ? 254 : biased));
}

// Packs the exact fp32 child boxes of a wide node (n of them) into the quantised node format. Along every axis the
// step is the smallest power of two with 254 steps covering the node, a lower corner is rounded down and an upper
// corner up to a step, and each rounded corner is then DECODED the way traversal decodes it (nodeCoord) and moved
// a further step outward if the decoded value does not bound the exact one: the decoded box contains the exact box.
__device__ BvhNode packNode(int n, const int32_t* c, const float (*lo)[3], const float (*hi)[3]) {
    BvhNode nd;
    nd.exps = 0;
    for (int j = 0; j < kNodeWidth; j++) nd.child[j] = j < n ? c[j] : kEmptyChild;
    for (int p = 0; p < 6; p++)
        for (int w = 0; w < kNodeQWords; w++) nd.q[p][w] = 0;
    for (uint32_t& x : nd.pad) x = 0;
    for (int a = 0; a < 3; a++) {
        float o = lo[0][a], top = hi[0][a];
        for (int j = 1; j < n; j++) { o = fminf(o, lo[j][a]); top = fmaxf(top, hi[j][a]); }
        nd.origin[a] = o;
        uint32_t e = stepExponent(o, top);
        for (;;) {  // raises the step in the (never observed) case that a corrected corner would not fit a byte
            const float step = nodeStep(e << (8 * a), a);
            bool fits = true;
            for (int j = 0; j < n && fits; j++) {
                int ql = 0, qh = 0;
                if (step > 0.0f) {
                    ql = (int)floorf((lo[j][a] - o) / step);
                    qh = (int)ceilf((hi[j][a] - o) / step);
                    if (ql < 0) ql = 0;
                    if (ql > 255) ql = 255;
                    if (qh < 0) qh = 0;
                }
                while (ql > 0 && nodeCoord(o, (uint32_t)ql, step) > lo[j][a]) ql--;
                while (qh <= 255 && nodeCoord(o, (uint32_t)qh, step) < hi[j][a]) qh++;
                if (qh > 255) { fits = false; break; }
                nd.q[a][j >> 2] = (nd.q[a][j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)ql << (8 * (j & 3)));
                nd.q[3 + a][j >> 2] = (nd.q[3 + a][j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)qh << (8 * (j & 3)));
            }
            if (fits || e >= 254) break;
            e++;
        }
        nd.exps |= e << (8 * a);
    }
    return nd;
}

// The full-precision twin of a wide node (small scenes only): the same children, fp32 boxes as they are.
__device__ ExactNode exactNode(int n, const int32_t* c, const float (*lo)[3], const float (*hi)[3]) {
    ExactNode nd;
    for (int j = 0; j < kNodeWidth; j++) {
        for (int k = 0; k < 3; k++) { nd.lo[j][k] = j < n ? lo[j][k] : 0.0f; nd.hi[j][k] = j < n ? hi[j][k] : 0.0f; }
        nd.child[j] = j < n ? c[j] : kEmptyChild;
    }
    for (int& x : nd.pad) x = 0;
    return nd;
}

__global__ void k_single_leaf(const FBox* leaf_boxes, BvhNode* nodes, ExactNode* xnodes) {
    int32_t c[1] = {~0};
    float lo[1][3], hi[1][3];
    for (int k = 0; k < 3; k++) { lo[0][k] = leaf_boxes[0].lo[k]; hi[0][k] = leaf_boxes[0].hi[k]; }
    nodes[0] = packNode(1, c, lo, hi);
    if (xnodes) xnodes[0] = exactNode(1, c, lo, hi);
}

// One top-down level of the collapse into wide nodes. A binary node marked with `level` becomes a wide node with
// the same index: starting from its two children, the inner child with the largest surface area is replaced by
// its own two children until the node holds kNodeWidth children or only leaves. The inner children that remain
// are marked for the next level; binary nodes that were swallowed leave their slots in the output array unused.
// mark[] starts as 0xffffffff except mark[root] = 0; a launch only reads marks an earlier launch wrote.
constexpr float kThinSteps = 8.0f;  // a leaf narrower than this many steps of its node gets a node of its own
__global__ void k_collapse_level(int n_internal, const BinNode* bin, uint32_t* mark, uint32_t level, BvhNode* out,
                                 ExactNode* xout, uint32_t* progress) {
    int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_internal) return;
    if (mark[i] != level) return;
    int32_t c[kNodeWidth];
    float lo[kNodeWidth][3], hi[kNodeWidth][3];
    const BinNode b = bin[i];
    for (int k = 0; k < 3; k++) { lo[0][k] = b.lo0[k]; hi[0][k] = b.hi0[k]; lo[1][k] = b.lo1[k]; hi[1][k] = b.hi1[k]; }
    c[0] = b.child0; c[1] = b.child1;
    int n = 2;
    while (n < kNodeWidth) {
        int pick = -1;
        float pick_area = -1.0f;
        for (int a = 0; a < n; a++) {
            if (c[a] < 0) continue;
            const float dx = hi[a][0] - lo[a][0], dy = hi[a][1] - lo[a][1], dz = hi[a][2] - lo[a][2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > pick_area) { pick_area = area; pick = a; }
        }
        if (pick < 0) break;
        const BinNode g = bin[c[pick]];
        for (int k = 0; k < 3; k++) { lo[pick][k] = g.lo0[k]; hi[pick][k] = g.hi0[k]; lo[n][k] = g.lo1[k]; hi[n][k] = g.hi1[k]; }
        c[pick] = g.child0; c[n] = g.child1;
        n++;
    }
    uint32_t inner = 0;
    for (int j = 0; j < n; j++)
        if (c[j] >= 0) { mark[c[j]] = level + 1; inner++; }
    if (xout) xout[i] = exactNode(n, c, lo, hi);
    // A leaf that is thin against the node's step along some axis (an axis-aligned wall beside a mesh: 2^-19 thick
    // in a node that spans the room) would come out of the quantisation many times thicker, and every ray that
    // starts or ends on the wall would run the fp64 test of its triangles. Such a leaf gets a node of its own
    // (slot n_internal + primitive index), whose step follows the leaf's own extent: a ray that only meets the
    // thickened box pays one more node step and never reaches the primitive test.
    float o[3], top[3];
    for (int a = 0; a < 3; a++) {
        o[a] = lo[0][a]; top[a] = hi[0][a];
        for (int j = 1; j < n; j++) { o[a] = fminf(o[a], lo[j][a]); top[a] = fmaxf(top[a], hi[j][a]); }
    }
    for (int j = 0; j < n; j++) {
        if (c[j] >= 0) continue;
        bool thin = false;
        for (int a = 0; a < 3; a++) {
            const float step = nodeStep(stepExponent(o[a], top[a]) << (8 * a), a);
            if (hi[j][a] - lo[j][a] < kThinSteps * step) thin = true;
        }
        if (!thin) continue;
        const int32_t slot = n_internal + ~c[j];
        out[slot] = packNode(1, &c[j], &lo[j], &hi[j]);
        c[j] = slot;
    }
    out[i] = packNode(n, c, lo, hi);
    if (inner) atomicAdd(progress, inner);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

}  // namespace

hipError_t buildLbvh(const DObject* d_objects, const Prim* d_prims_unsorted, uint32_t n, hipStream_t s,
                     BvhBuildResult* out) {
    hipError_t err = hipSuccess;
    out->nodes = nullptr; out->xnodes = nullptr; out->prims = nullptr; out->num_nodes = 0;
    for (int k = 0; k < 3; k++) { out->scene_lo[k] = 0; out->scene_hi[k] = 0; }
    if (n == 0) return hipSuccess;

    Box6* boxes = nullptr; FBox* leaf_boxes = nullptr; FBox* node_boxes = nullptr;
    unsigned long long *scene = nullptr, *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *idx = nullptr, *idx_sorted = nullptr, *ready = nullptr, *progress = nullptr, *depth = nullptr;
    BinNode* bin = nullptr;
    int32_t *child0 = nullptr, *child1 = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_bytes = 0;
    BvhNode* nodes = nullptr; Prim* prims = nullptr;
    ExactNode* xnodes = nullptr;
    uint32_t n_internal = n > 1 ? n - 1 : 1;
    unsigned g = (n + kB - 1) / kB;
    unsigned long long scene_init[6];
    unsigned long long h_scene[6];

    CK(hipMalloc(&boxes, sizeof(Box6) * n));
    CK(hipMalloc(&leaf_boxes, sizeof(FBox) * n));
    CK(hipMalloc(&node_boxes, sizeof(FBox) * n_internal));
    CK(hipMalloc(&scene, sizeof(unsigned long long) * 6));
    CK(hipMalloc(&keys, sizeof(unsigned long long) * n));
    CK(hipMalloc(&keys_sorted, sizeof(unsigned long long) * n));
    CK(hipMalloc(&idx, sizeof(uint32_t) * n));
    CK(hipMalloc(&idx_sorted, sizeof(uint32_t) * n));
    CK(hipMalloc(&ready, sizeof(uint32_t) * n_internal));
    CK(hipMalloc(&progress, sizeof(uint32_t)));
    CK(hipMalloc(&child0, sizeof(int32_t) * n_internal));
    CK(hipMalloc(&child1, sizeof(int32_t) * n_internal));
    CK(hipMalloc(&nodes, sizeof(BvhNode) * ((size_t)n_internal + n)));  // + one slot per primitive for leaves that get a node of their own
    if (n <= kExactTreePrims) CK(hipMalloc(&xnodes, sizeof(ExactNode) * n_internal));
    CK(hipMalloc(&bin, sizeof(BinNode) * n_internal));
    CK(hipMalloc(&depth, sizeof(uint32_t) * n_internal));
    CK(hipMalloc(&prims, sizeof(Prim) * n));

    for (int k = 0; k < 3; k++) { scene_init[k] = ~0ULL; scene_init[3 + k] = 0ULL; }
    CK(hipMemcpyAsync(scene, scene_init, sizeof(scene_init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_prim_bounds, dim3(g), dim3(kB), 0, s, d_objects, d_prims_unsorted, n, boxes, scene);
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(kB), 0, s, boxes, n, scene, keys, idx);
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys, keys_sorted, idx, idx_sorted, (int)n, 0, 63, s));
    CK(hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 16));
    CK(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys, keys_sorted, idx, idx_sorted, (int)n, 0, 63, s));
    hipLaunchKernelGGL(k_gather, dim3(g), dim3(kB), 0, s, d_prims_unsorted, boxes, idx_sorted, n, scene, prims, leaf_boxes);
    if (n == 1) {
        hipLaunchKernelGGL(k_single_leaf, dim3(1), dim3(1), 0, s, leaf_boxes, nodes, xnodes);
    } else {
        unsigned gi = (n_internal + kB - 1) / kB;
        // binary hierarchy over the sorted primitives: clustering along the Morton order (default), or the Karras radix
        // tree (UBDPT_BVH_BUILDER=karras; also what is used should the clustering fail to finish)
        static const bool use_karras = [] { const char* e = getenv("UBDPT_BVH_BUILDER"); return e && !strcmp(e, "karras"); }();
        static const int ploc_radius = [] { const char* e = getenv("UBDPT_PLOC_RADIUS"); int r = e ? atoi(e) : kPlocRadius; return r < 1 ? 1 : (r > kPlocMaxRadius ? kPlocMaxRadius : r); }();
        bool clustered = false;
        if (!use_karras && n >= 2) {
            FBox* cbox[2] = {nullptr, nullptr};
            int32_t* cid[2] = {nullptr, nullptr};
            uint32_t* nn = nullptr;
            unsigned long long *flags = nullptr, *scan = nullptr;
            void* scan_tmp = nullptr;
            size_t scan_bytes = 0;
            hipError_t pe = hipSuccess;
            auto pck = [&](hipError_t e) { if (e != hipSuccess && pe == hipSuccess) pe = e; };
            for (int k = 0; k < 2; k++) { pck(hipMalloc(&cbox[k], sizeof(FBox) * n)); pck(hipMalloc(&cid[k], sizeof(int32_t) * n)); }
            pck(hipMalloc(&nn, sizeof(uint32_t) * n));
            pck(hipMalloc(&flags, sizeof(unsigned long long) * n));
            pck(hipMalloc(&scan, sizeof(unsigned long long) * n));
            if (pe == hipSuccess) pck(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags, scan, (int)n, s));
            if (pe == hipSuccess) pck(hipMalloc(&scan_tmp, scan_bytes ? scan_bytes : 16));
            if (pe == hipSuccess) {
                pck(hipMemcpyAsync(cbox[0], leaf_boxes, sizeof(FBox) * n, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(k_ploc_init, dim3(g), dim3(kB), 0, s, n, cid[0]);
                uint32_t m = n, id_top = n_internal;
                int cur = 0;
                for (int round = 0; round < 1024 && m > 1 && pe == hipSuccess; round++) {
                    const unsigned gm = (m + kB - 1) / kB;
                    hipLaunchKernelGGL(k_ploc_nearest, dim3(gm), dim3(kB), 0, s, cbox[cur], m, nn, ploc_radius);
                    hipLaunchKernelGGL(k_ploc_flags, dim3(gm), dim3(kB), 0, s, nn, m, flags);
                    pck(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, flags, scan, (int)m, s));
                    hipLaunchKernelGGL(k_ploc_merge, dim3(gm), dim3(kB), 0, s, nn, flags, scan, m, cbox[cur], cid[cur],
                                       cbox[cur ^ 1], cid[cur ^ 1], id_top, child0, child1);
                    unsigned long long last_scan = 0, last_flag = 0;
                    pck(hipMemcpyAsync(&last_scan, scan + (m - 1), sizeof(last_scan), hipMemcpyDeviceToHost, s));
                    pck(hipMemcpyAsync(&last_flag, flags + (m - 1), sizeof(last_flag), hipMemcpyDeviceToHost, s));
                    pck(hipStreamSynchronize(s));
                    const unsigned long long tot = last_scan + last_flag;
                    const uint32_t kept = (uint32_t)(tot & 0xffffffffULL), merged = (uint32_t)(tot >> 32);
                    if (merged == 0 || kept + merged != m) break;  // cannot happen (the smallest pair in reach is mutual)
                    m = kept;
                    id_top -= merged;
                    cur ^= 1;
                }
                clustered = pe == hipSuccess && m == 1 && id_top == 0;
            }
            for (int k = 0; k < 2; k++) { hipFree(cbox[k]); hipFree(cid[k]); }
            hipFree(nn); hipFree(flags); hipFree(scan); hipFree(scan_tmp);
            if (pe != hipSuccess) { err = pe; goto done; }
        }
        if (!clustered)
            hipLaunchKernelGGL(k_hierarchy, dim3(gi), dim3(kB), 0, s, keys_sorted, (int)n, child0, child1);
        CK(hipMemsetAsync(ready, 0, sizeof(uint32_t) * n_internal, s));
        CK(hipMemsetAsync(progress, 0, sizeof(uint32_t), s));
        uint32_t done_nodes = 0, level = 1;
        // tree depth is bounded by the key length plus the index tie-break bits
        while (done_nodes < n_internal && level <= 4096) {
            for (int rep = 0; rep < 8; rep++, level++)
                hipLaunchKernelGGL(k_refit_level, dim3(gi), dim3(kB), 0, s, (int)n_internal, child0, child1, leaf_boxes,
                                   node_boxes, ready, level, bin, progress);
            CK(hipMemcpyAsync(&done_nodes, progress, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
        }
        if (done_nodes < n_internal) { err = hipErrorUnknown; goto done; }
        // collapse into wide nodes, top-down, one launch per level of the wide tree; done when a round of launches
        // marks no further node
        {
            const uint32_t zero = 0;
            CK(hipMemsetAsync(depth, 0xff, sizeof(uint32_t) * n_internal, s));
            CK(hipMemcpyAsync(depth, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, s));
            CK(hipMemsetAsync(progress, 0, sizeof(uint32_t), s));
            uint32_t seen = 0, before = 0xffffffffu, lv = 0;
            while (seen != before && lv <= 4096) {
                before = seen;
                for (int rep = 0; rep < 8; rep++, lv++)
                    hipLaunchKernelGGL(k_collapse_level, dim3(gi), dim3(kB), 0, s, (int)n_internal, bin, depth, lv, nodes,
                                       xnodes, progress);
                CK(hipMemcpyAsync(&seen, progress, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
            }
            if (seen != before) { err = hipErrorUnknown; goto done; }
        }
    }
    CK(hipMemcpyAsync(h_scene, scene, sizeof(h_scene), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    for (int k = 0; k < 3; k++) {
        unsigned long long b = h_scene[k];
        b = (b & 0x8000000000000000ULL) ? (b & 0x7fffffffffffffffULL) : ~b;
        double lo; memcpy(&lo, &b, 8);
        b = h_scene[3 + k];
        b = (b & 0x8000000000000000ULL) ? (b & 0x7fffffffffffffffULL) : ~b;
        double hi; memcpy(&hi, &b, 8);
        out->scene_lo[k] = (float)lo; out->scene_hi[k] = (float)hi;
    }
    out->nodes = nodes; out->xnodes = xnodes; out->prims = prims; out->num_nodes = n_internal + n;
    nodes = nullptr; xnodes = nullptr; prims = nullptr;
done:
    hipFree(boxes); hipFree(leaf_boxes); hipFree(node_boxes); hipFree(scene); hipFree(keys); hipFree(keys_sorted);
    hipFree(idx); hipFree(idx_sorted); hipFree(ready); hipFree(progress); hipFree(child0); hipFree(child1);
    hipFree(sort_tmp); hipFree(nodes); hipFree(xnodes); hipFree(prims); hipFree(bin); hipFree(depth);
    return err;
}

}  // namespace ub
