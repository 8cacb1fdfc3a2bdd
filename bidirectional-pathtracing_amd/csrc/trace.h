// trace.h — ray / scene intersection on the device.
//
// Replaces UScene::intersectionPoint and UScene::visibility (uengine/uscene.cpp:21-66), which scan
// every object and every face. Here a ray walks a world-space LBVH (wide nodes with quantised child boxes,
// scene.h) with a per-lane stack whose first kLdsStack entries live in LDS ([entry][lane], bank = lane, conflict
// free) and whose rare overflow goes to a per-lane slice of HBM. Boxes are tested in fp32 with a
// conservative slab test; candidate primitives are tested
// exactly as the reference tests them: in object-local space, in fp64, in the reference's operation
// order (URay::intersectTriangle / intersectUnitSphere, uengine/ugeometry.cpp:70-114), including
// the per-mesh bounding-sphere gate (uapp/mesh.cpp:120,167). The reference's "first object, first
// face wins ties" rule (strict '<' in uscene.cpp:52 and mesh.cpp:128) becomes the lexicographic
// minimum of (d, object, face), which is independent of traversal order.
#pragma once

#include "scene.h"

namespace ub {

#define DEV __device__ __forceinline__

#ifndef UBDPT_LDS_STACK
#define UBDPT_LDS_STACK 16
#endif
// stack entries per lane kept in LDS (16 KB per workgroup; 12 were 1.5 % slower on the 2 M-triangle scene, 14 to 24 the same)
constexpr int kLdsStack = UBDPT_LDS_STACK;
constexpr int kOverflowStack = 96; // further entries per lane in HBM

struct Ray {
    d3 o, d;  // URay origin (w = 1) and direction (w = 0); the direction is NOT renormalised
};

// ugeometry.cpp:70-94
DEV bool intersectUnitSphere(d3 o, d3 dir, double& d) {
    double a = dot(dir, dir);
    double b = 2 * dot(o, dir);
    double c = dot(o, o) - 1;
    double delta = (b * b) - (4 * a * c);
    if (delta < 0) return false;
    double sd = sqrt(delta);
    d = (-b - sd) / (2 * a);
    if (d > 0) return true;
    d = (-b + sd) / (2 * a);
    if (d < 0) return false;
    return true;
}

// ugeometry.cpp:96-114
DEV bool intersectTriangle(d3 o, d3 dir, d3 p0, d3 p1, d3 p2, double& d, double& u, double& v) {
    d3 e1 = p1 - p0;
    d3 e2 = p2 - p0;
    d3 m = o - p0;
    d3 c1 = cross(dir, e2);
    d3 c2 = cross(m, e1);
    double a = dot(e1, c1);
    d = dot(e2, c2) / a;
    u = dot(m, c1) / a;
    v = dot(dir, c2) / a;
    return d > 0 && u >= 0 && v >= 0 && u + v <= 1;
}

// URay::transform (ugeometry.cpp:55-58)
DEV Ray toLocal(const m4& invW, const Ray& r) {
    Ray l;
    l.o = mul_xyz(invW, r.o, 1.0);
    l.d = mul_xyz(invW, r.d, 0.0);
    return l;
}

DEV d3 loadP(const float* p) { return mk3((double)p[0], (double)p[1], (double)p[2]); }

// the closest candidate so far
struct Closest {
    double d, u, v;
    int32_t object;   // -1: none
    uint32_t face;
};

struct TraceCounters {
    uint32_t nodes, prims;            // per lane: node visits, primitive tests
    uint32_t wave_nodes, wave_prims;  // per wave: times the wave executed a node step / a primitive test
    uint32_t max_stack;               // deepest traversal stack seen (entries)
};
// Counts one wave-level execution of the code it stands in: nodes / (64 * wave_nodes) is the SIMD
// utilisation of the node step, prims / (64 * wave_prims) that of the primitive test.
DEV void countWaveStep(uint32_t& c) {
    unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)m) - 1)) c++;
}

// Per-ray cache of the object-local ray and of the bounding-sphere gate result.
struct LocalCache {
    Ray l;
    int32_t object;       // object the local ray belongs to (-1: none)
    int32_t gate_object;  // object the gate was evaluated for
    bool gate_ok;
};

DEV void cacheObject(const DScene& S, const Ray& r, int32_t obj, LocalCache& c) {
    if (c.object != obj) {
        c.l = toLocal(S.objects[obj].invW, r);
        c.object = obj;
    }
}
// rayL.transform(m_bounding_sphere).intersectUnitSphere(d)  (uapp/mesh.cpp:120,167)
DEV bool meshGate(const DScene& S, int32_t obj, LocalCache& c) {
    if (c.gate_object != obj) {
        Ray g = toLocal(S.objects[obj].bsphere, c.l);
        double d;
        c.gate_ok = intersectUnitSphere(g.o, g.d, d);
        c.gate_object = obj;
    }
    return c.gate_ok;
}

DEV bool lexLess(double d, int32_t obj, uint32_t face, const Closest& b) {
    return d < b.d || (d == b.d && (obj < b.object || (obj == b.object && face < b.face)));
}

// Tests one primitive for the closest-hit query.
DEV void testPrimClosest(const DScene& S, const Ray& r, const Prim& pr, LocalCache& c, Closest& best) {
    int32_t obj = (int32_t)pr.object;
    cacheObject(S, r, obj, c);
    double d, u = 0, v = 0;
    if (pr.face == kSphereFace) {
        if (!intersectUnitSphere(c.l.o, c.l.d, d)) return;
    } else {
        if (!intersectTriangle(c.l.o, c.l.d, loadP(pr.p), loadP(pr.p + 3), loadP(pr.p + 6), d, u, v)) return;
    }
    if (!lexLess(d, obj, pr.face, best)) return;
    if (pr.face != kSphereFace && !meshGate(S, obj, c)) return;
    best.d = d; best.u = u; best.v = v; best.object = obj; best.face = pr.face;
}

// Tests one primitive for the visibility query: does it block (0 < d < dist)?
DEV bool testPrimBlocks(const DScene& S, const Ray& r, const Prim& pr, LocalCache& c, double dist) {
    int32_t obj = (int32_t)pr.object;
    cacheObject(S, r, obj, c);
    double d, u, v;
    if (pr.face == kSphereFace) {
        if (!intersectUnitSphere(c.l.o, c.l.d, d)) return false;
    } else {
        if (!intersectTriangle(c.l.o, c.l.d, loadP(pr.p), loadP(pr.p + 3), loadP(pr.p + 6), d, u, v)) return false;
    }
    if (!((d < dist) && (d > 0))) return false;
    if (pr.face != kSphereFace && !meshGate(S, obj, c)) return false;
    return true;
}

// ---- conservative fp32 slab test -----------------------------------------------------------------
// The tree is walked in fp32 (half the VALU cost and registers of fp64); only the primitive tests
// are fp64. The fp32 ray (origin and direction rounded to float) deviates from the fp64 ray by at
// most 2^-24 (2|o| + M) inside a scene of magnitude M; boxes are padded by 2^-20 M at build time
// (bvh.hip) and by `pad` = 2^-22 |o| per ray here, and the slab interval is widened by 2^-21
// relative (more than the 2*gamma_3 bound of robust fp32 slab arithmetic), so a box is never
// rejected for a ray parameter at which the fp64 primitive test can accept.
struct RayF {
    float ox, oy, oz;
    float ix, iy, iz;  // 1 / direction
    float pad;         // origin-magnitude dependent extra box padding
};
DEV RayF makeRayF(const Ray& r) {
    RayF q;
    q.ox = (float)r.o.x; q.oy = (float)r.o.y; q.oz = (float)r.o.z;
    // 1 / direction, limited to +-2^64: a ray parallel to a slab gets entry and exit parameters of +-huge or 0 (origin
    // on the plane) exactly as an infinite reciprocal would give, never a NaN from 0 * inf, and every product stays finite
    const float big = 18446744073709551616.0f;
    q.ix = fminf(fmaxf(1.0f / (float)r.d.x, -big), big);
    q.iy = fminf(fmaxf(1.0f / (float)r.d.y, -big), big);
    q.iz = fminf(fmaxf(1.0f / (float)r.d.z, -big), big);
    q.pad = fmaxf(fabsf(q.ox), fmaxf(fabsf(q.oy), fabsf(q.oz))) * 2.384185791015625e-07f;  // 2^-22
    return q;
}
DEV bool slabF(const float* lo, const float* hi, const RayF& q, float tmax, float& tnear) {
    float t0 = ((lo[0] - q.pad) - q.ox) * q.ix, t1 = ((hi[0] + q.pad) - q.ox) * q.ix;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = ((lo[1] - q.pad) - q.oy) * q.iy; t1 = ((hi[1] + q.pad) - q.oy) * q.iy;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    t0 = ((lo[2] - q.pad) - q.oz) * q.iz; t1 = ((hi[2] + q.pad) - q.oz) * q.iz;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    const float k = 4.76837158203125e-07f;  // 2^-21
    tn = tn - fabsf(tn) * k;
    tf = tf + fabsf(tf) * k;
    tnear = tn;
    return tn <= tf && tf >= 0.0f && tn <= tmax;
}
// smallest float >= x (x > 0 or +inf)
DEV float floatUp(double x) { return __double2float_ru(x); }

// Per-lane traversal stack: LDS for the first kLdsStack entries, HBM beyond.
struct Stack {
    int32_t* lds;       // &lds_stack[0][lane_in_block], stride = blockDim.x
    int32_t* ovf;       // &overflow[0][global_lane], stride = ovf_stride
    uint32_t lds_stride;
    uint64_t ovf_stride;
    int sp;
    int cap;            // entries this lane may use, at most kLdsStack + kOverflowStack
    bool overflowed;    // a push found the stack full: the subtree was dropped, the result is not to be trusted
    DEV void push(int32_t v) {
        // (two separate stores: a store through a pointer selected between LDS and device memory does not compile)
        const int lim = cap < kLdsStack ? cap : kLdsStack;
        if (sp < lim) lds[sp * lds_stride] = v;
        else if (sp < cap) ovf[(uint64_t)(sp - kLdsStack) * ovf_stride] = v;
        else { overflowed = true; return; }
        sp++;
    }
    DEV int32_t pop() {
        sp--;
        return sp < kLdsStack ? lds[sp * lds_stride] : ovf[(uint64_t)(sp - kLdsStack) * ovf_stride];
    }
};

constexpr int32_t kTraversalDone = kEmptyChild;

// One traversal step from an internal node: tests the children, returns the nearest one that is hit and pushes
// the others (farthest first), or pops / finishes when no child is hit.
DEV void sortPair(float& ta, int32_t& ca, float& tb, int32_t& cb) {
    if (tb < ta) { float t = ta; ta = tb; tb = t; int32_t c = ca; ca = cb; cb = c; }
}
DEV BvhNode loadNode(const DScene& S, int32_t node) {
    return *reinterpret_cast<const BvhNode*>(reinterpret_cast<const char*>(S.nodes) + (size_t)node * S.node_stride);
}
// entry distances of the children, near to far, then: the nearest child that is hit is returned, the others pushed
DEV int32_t sortAndPush(float (&t)[kNodeWidth], int32_t (&c)[kNodeWidth], Stack& st, TraceCounters* tc) {
    const float inf = __builtin_inff();
    // entry distances are finite for a hit (the ray parameter interval is clipped to [.., tmax]); sort near to far
    // with a comparator network (5 comparators for four, 19 for eight), misses (inf) end up last
    if (kNodeWidth == 4) {
        sortPair(t[0], c[0], t[1], c[1]); sortPair(t[2], c[2], t[3], c[3]);
        sortPair(t[0], c[0], t[2], c[2]); sortPair(t[1], c[1], t[3], c[3]);
        sortPair(t[1], c[1], t[2], c[2]);
    } else {
        constexpr int kNet[19][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {1, 2}, {5, 6},
                                     {0, 4}, {3, 7}, {1, 5}, {2, 6}, {1, 4}, {3, 6}, {2, 4}, {3, 5}, {3, 4}};
#pragma unroll
        for (int k = 0; k < 19; k++) sortPair(t[kNet[k][0] % kNodeWidth], c[kNet[k][0] % kNodeWidth], t[kNet[k][1] % kNodeWidth], c[kNet[k][1] % kNodeWidth]);
    }
#pragma unroll
    for (int k = kNodeWidth - 1; k >= 1; k--)
        if (t[k] < inf) st.push(c[k]);
    if (tc && (uint32_t)st.sp > tc->max_stack) tc->max_stack = (uint32_t)st.sp;
    if (t[0] < inf) return c[0];
    return st.sp ? st.pop() : kTraversalDone;
}
// Slab tests of the children of a quantised node without decoding their boxes. Along axis a a child plane sits at
// origin + byte * step; its ray parameter is computed as fma(byte, step, base) * (1 / d), where base =
// (origin - o) -+ pad is formed once per node. Which plane a ray enters a slab through is known from the sign of its
// direction, so the near and far planes are picked per NODE (the bytes of all children for one plane are one word)
// and a child costs 6 conversions, 6 fused multiply-adds, 6 multiplications, a max3 and a min3.
// Conservative: against the real-number plane origin + byte * step (which the builder holds within half an ulp of
// the exact box, bvh.hip) the position fma(...) is off by the roundings of origin - o, of base and of the fma
// itself: at most 3 * 2^-24 (M + |o|) for a scene of magnitude M; the M part, like the rounding of the ray itself
// (2^-24 (2|o| + M), above), is inside the 2^-20 M every box was padded with at build time, the |o| part inside the
// per-ray pad, which is doubled here to 2^-21 |o|; the product with 1 / d and its rounding are covered by the 2^-21
// relative widening of the interval, as in slabF.
DEV int32_t descendNode(const BvhNode& nd, const RayF& q, float tmax, Stack& st, TraceCounters* tc) {
    if (tc) { tc->nodes++; countWaveStep(tc->wave_nodes); }
    const float inf = __builtin_inff();
    const float kWiden = 4.76837158203125e-07f;  // 2^-21
    const float pad2 = q.pad + q.pad;
    const float o[3] = {q.ox, q.oy, q.oz}, iv[3] = {q.ix, q.iy, q.iz};
    float step[3], nbase[3], fbase[3];
    uint32_t qn[3][kNodeQWords], qf[3][kNodeQWords];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        step[a] = nodeStep(nd.exps, a);
        const float dmo = nd.origin[a] - o[a];
        const float lb = dmo - pad2, hb = dmo + pad2;
        const bool neg = iv[a] < 0.0f;
        nbase[a] = neg ? hb : lb;
        fbase[a] = neg ? lb : hb;
#pragma unroll
        for (int w = 0; w < kNodeQWords; w++) {
            qn[a][w] = neg ? nd.q[3 + a][w] : nd.q[a][w];
            qf[a][w] = neg ? nd.q[a][w] : nd.q[3 + a][w];
        }
    }
    float t[kNodeWidth];
    int32_t c[kNodeWidth];
#pragma unroll
    for (int k = 0; k < kNodeWidth; k++) {
        c[k] = nd.child[k];
        const int w = k >> 2, sh = 8 * (k & 3);
        float tn = -inf, tf = inf;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float xn = __builtin_fmaf((float)((qn[a][w] >> sh) & 0xffu), step[a], nbase[a]);
            const float xf = __builtin_fmaf((float)((qf[a][w] >> sh) & 0xffu), step[a], fbase[a]);
            tn = fmaxf(tn, xn * iv[a]);
            tf = fminf(tf, xf * iv[a]);
        }
        tn = tn - fabsf(tn) * kWiden;
        tf = tf + fabsf(tf) * kWiden;
        t[k] = (c[k] != kEmptyChild && tn <= tf && tf >= 0.0f && tn <= tmax) ? tn : inf;
    }
    return sortAndPush(t, c, st, tc);
}
DEV int32_t descendExact(const ExactNode& nd, const RayF& q, float tmax, Stack& st, TraceCounters* tc) {
    if (tc) { tc->nodes++; countWaveStep(tc->wave_nodes); }
    const float inf = __builtin_inff();
    float t[kNodeWidth];
    int32_t c[kNodeWidth];
#pragma unroll
    for (int k = 0; k < kNodeWidth; k++) {
        c[k] = nd.child[k];
        if (!(c[k] != kEmptyChild && slabF(nd.lo[k], nd.hi[k], q, tmax, t[k]))) t[k] = inf;
    }
    return sortAndPush(t, c, st, tc);
}
// EXACT: walk the full-precision twin of the tree (S.xnodes)
template <bool EXACT>
DEV int32_t descendIn(const DScene& S, int32_t node, const RayF& q, float tmax, Stack& st, TraceCounters* tc) {
    if (EXACT)
        return descendExact(*reinterpret_cast<const ExactNode*>(reinterpret_cast<const char*>(S.xnodes) + (size_t)node * S.xnode_stride), q, tmax, st, tc);
    return descendNode(loadNode(S, node), q, tmax, st, tc);
}

// UScene::intersectionPoint through the LBVH, one ray per lane. Returns the lexicographically smallest
// (d, object, face) (object -1: no hit). while-while form: every lane of the wave descends until it holds a leaf
// (or is done), then the fp64 primitive tests of all lanes run together. (The kernels that trace the rays of
// large scenes walk the same tree with lanes refilled from a queue: wavefront_split.h.)
template <bool EXACT = false>
DEV Closest closestHitBvh(const DScene& S, const Ray& r, Stack& st, TraceCounters* tc) {
    Closest best;
    best.d = __builtin_inf(); best.u = 0; best.v = 0; best.object = -1; best.face = 0;
    if (S.num_prims == 0) return best;
    LocalCache c;
    c.object = -1; c.gate_object = -1; c.gate_ok = false;
    const RayF q = makeRayF(r);
    float tmax = floatUp(best.d);
    st.sp = 0;
    int32_t node = 0;
    for (;;) {
        while (node >= 0 && node != kTraversalDone) node = descendIn<EXACT>(S, node, q, tmax, st, tc);
        if (node == kTraversalDone) break;
        if (tc) { tc->prims++; countWaveStep(tc->wave_prims); }
        testPrimClosest(S, r, S.prims[~node], c, best);
        tmax = floatUp(best.d);
        node = st.sp ? st.pop() : kTraversalDone;
    }
    return best;
}

// UScene::visibility through the LBVH: true when nothing blocks p0 -> p1 (the ray uscene.cpp:21-27 builds).
template <bool EXACT = false>
DEV bool visibleBvh(const DScene& S, d3 p0, d3 p1, Stack& st, TraceCounters* tc) {
    double dist = length(p1 - p0);
    Ray r;
    r.o = p0;
    r.d = (p1 - p0) / dist;
    if (S.num_prims == 0) return true;
    LocalCache c;
    c.object = -1; c.gate_object = -1; c.gate_ok = false;
    const RayF q = makeRayF(r);
    const float tmax = floatUp(dist);
    st.sp = 0;
    int32_t node = 0;
    for (;;) {
        while (node >= 0 && node != kTraversalDone) node = descendIn<EXACT>(S, node, q, tmax, st, tc);
        if (node == kTraversalDone) break;
        if (tc) { tc->prims++; countWaveStep(tc->wave_prims); }
        if (testPrimBlocks(S, r, S.prims[~node], c, dist)) return false;
        node = st.sp ? st.pop() : kTraversalDone;
    }
    return true;
}

// ---- linear-scan accelerator: the reference's own loops, for cross-checking the tree --------------
DEV Closest closestHitLinear(const DScene& S, const Ray& r, TraceCounters* tc) {
    Closest best;
    best.d = __builtin_inf(); best.u = 0; best.v = 0; best.object = -1; best.face = 0;
    for (uint32_t o = 0; o < S.num_objects; o++) {
        const DObject& ob = S.objects[o];
        Ray l = toLocal(ob.invW, r);
        double d, u, v;
        if (ob.kind == 0) {  // implicit sphere, uapp/implicitsphere.cpp:3-6
            if (tc) tc->prims++;
            if (intersectUnitSphere(l.o, l.d, d) && d < best.d) {
                best.d = d; best.u = 0; best.v = 0; best.object = (int32_t)o; best.face = kSphereFace;
            }
        } else {  // uapp/mesh.cpp:108-142
            Ray g = toLocal(ob.bsphere, l);
            if (!intersectUnitSphere(g.o, g.d, d)) continue;
            double cd = __builtin_inf(), cu = 0, cv = 0;
            uint32_t cf = 0;
            bool hit = false;
            for (uint64_t f = 0; f < ob.num_faces; f++) {
                const float* p = S.face_pos + 9 * (ob.first_face + f);
                if (tc) tc->prims++;
                if (intersectTriangle(l.o, l.d, loadP(p), loadP(p + 3), loadP(p + 6), d, u, v) && d < cd) {
                    cd = d; cu = u; cv = v; cf = (uint32_t)f; hit = true;
                }
            }
            if (hit && cd < best.d) {
                best.d = cd; best.u = cu; best.v = cv; best.object = (int32_t)o; best.face = cf;
            }
        }
    }
    return best;
}

// `r` = the ray UScene::visibility builds from its two points (uscene.cpp:24-27), dist = their distance
DEV bool visibleLinearRay(const DScene& S, const Ray& r, double dist, TraceCounters* tc) {
    for (uint32_t o = 0; o < S.num_objects; o++) {
        const DObject& ob = S.objects[o];
        Ray l = toLocal(ob.invW, r);
        double d, u, v;
        if (ob.kind == 0) {
            if (tc) tc->prims++;
            if (intersectUnitSphere(l.o, l.d, d) && (d < dist) && (d > 0)) return false;
        } else {  // uapp/mesh.cpp:161-189
            Ray g = toLocal(ob.bsphere, l);
            if (!intersectUnitSphere(g.o, g.d, d)) continue;
            double md = __builtin_inf();
            bool hit = false;
            for (uint64_t f = 0; f < ob.num_faces; f++) {
                const float* p = S.face_pos + 9 * (ob.first_face + f);
                if (tc) tc->prims++;
                if (intersectTriangle(l.o, l.d, loadP(p), loadP(p + 3), loadP(p + 6), d, u, v) && d < md) {
                    md = d; hit = true;
                }
            }
            if (hit && (md < dist) && (md > 0)) return false;
        }
    }
    return true;
}
DEV bool visibleLinear(const DScene& S, d3 p0, d3 p1, TraceCounters* tc) {
    double dist = length(p1 - p0);
    Ray r;
    r.o = p0;
    r.d = (p1 - p0) / dist;
    return visibleLinearRay(S, r, dist, tc);
}

// ---- surface point of the winning candidate, object-local space ------------------------------------
struct SurfacePoint {  // USurfacePoint (uengine/uutils.h:22-35) without the matrices
    d3 pos, Ng, Ns, Ts, Bs;
    double tex_u, tex_v;
    int32_t object;
    int32_t lobe;
    uint32_t face;
};

DEV void surfacePoint(const DScene& S, const Ray& r, const Closest& h, SurfacePoint& sp) {
    const DObject& ob = S.objects[h.object];
    Ray l = toLocal(ob.invW, r);
    sp.object = h.object;
    sp.face = h.face;
    sp.pos = l.o + h.d * l.d;
    if (h.face == kSphereFace) {  // uapp/implicitsphere.cpp:8-15
        sp.tex_u = 0; sp.tex_v = 0;
        sp.Ns = sp.Ng = normalize(sp.pos);
        sp.Ts = normalize(-sp.Ns + mk3(0, 0, 1.0 / sp.Ns.z));
        sp.Bs = normalize(cross(sp.Ns, sp.Ts));
    } else {  // uapp/mesh.cpp:144-156
        const FaceAttr& fa = S.face_attr[ob.first_face + h.face];
        double w0 = 1.0 - h.u - h.v;
        sp.tex_u = w0 * (double)fa.uv[0] + h.u * (double)fa.uv[2] + h.v * (double)fa.uv[4];
        sp.tex_v = w0 * (double)fa.uv[1] + h.u * (double)fa.uv[3] + h.v * (double)fa.uv[5];
        sp.Ns = normalize(w0 * loadP(fa.n) + h.u * loadP(fa.n + 3) + h.v * loadP(fa.n + 6));
        sp.Ng = mk3(fa.Ng[0], fa.Ng[1], fa.Ng[2]);
        if (dot(sp.Ns, sp.Ng) < 0) sp.Ng = sp.Ng * -1.0;
        sp.Ts = normalize(w0 * loadP(fa.t) + h.u * loadP(fa.t + 3) + h.v * loadP(fa.t + 6));
        sp.Bs = normalize(cross(sp.Ns, sp.Ts));
    }
}

}  // namespace ub
