// paths.h — path-vertex storage in HBM and the BDPT measurement routines that read it.
//
// Vertex storage is an array of 192-byte records (24 doubles, three 64-byte lines). Subpaths die at
// different depths, so a layout with one plane per field would leave the planes of the deeper slots
// sparsely used, and every 8-byte access of a lone survivor would still fetch a whole 64-byte line
// from HBM; with one record per vertex every fetched line is data of the vertex that asked for it.
// The fields are grouped by consumer: line 0 holds what the ray tracing stages and the MIS walks
// read (position, the two area pdfs, object / flags), line 1 the normals, line 2 the rest of the
// shading frame and the throughput (moving the geometric normal into line 0 and the specular flags
// into the subpath's count word was measured: the visibility kernel gained 6 %, evaluation lost
// more). Instead of its texture coordinates a vertex stores the texture's colour there, which
// shading has looked up anyway: a connection then evaluates the BSDF without following
// object -> texture -> texel. A vertex is UPathVertex
// (uengine/ubdptrenderer.h:9-24) minus the two 128-byte matrices and the shared_ptrs, which are
// replaced by the object index and the lobe id.
#pragma once

#include "bsdf.h"

namespace ub {

enum { F_POS = 0, F_PL = 3, F_PE = 4, F_OBJFLAGS = 5, F_TEX = 6, F_NG = 9, F_NS = 12, F_TS = 15, F_BS = 18, F_A = 21,
       kVertexDoubles = 24 };

// The count word of a subpath: bits 0..7 = vertices stored, bit 8 + k = vertex k is specular
// (k < kMaskedVertices; later vertices are looked up in their records).
constexpr uint32_t kMaskedVertices = 24;
DEV uint32_t specularBit(uint32_t k, bool specular) { return (specular && k < kMaskedVertices) ? (1u << (8 + k)) : 0u; }

// The first `dense` slots of every subpath are stored densely ([slot][path]). Few subpaths get past them (on box scenes
// of diffuse and glass surfaces 0.4 to 1.6 % reach slot 8, 1 in 10 000 slot 16), so the slots from `dense` on are not
// laid out for every subpath: they live in pages of kPageSlots consecutive slots of ONE subpath, handed out from a pool
// (one atomic) when a subpath gets there, and found through a page table ([level][path], 4 bytes per entry). A batch whose
// subpaths need more pages than the pool holds cannot be completed: the subpath that finds the pool empty ends there, a
// counter is raised, and the call that rendered the batch fails (see api.cpp), as it does for a full traversal stack.
constexpr uint32_t kPageSlots = 8;

// PAGED = false: a view of the same store for code that only touches the dense slots (the connection evaluation of
// pairs whose end vertices lie in them: its record addresses are then plain arithmetic, which keeps that kernel at two
// waves per SIMD; with the page lookup in every address it needs all 512 registers of a lane).
template <bool PAGED>
struct PathStoreT {
    double* v;        // records of kVertexDoubles doubles: the dense slots, then the pool's pages
    uint32_t* count;  // count word of subpath i at count[i * count_stride]
    uint64_t n;       // subpaths of the batch (stride of a dense slot)
    uint32_t maxv;
    uint32_t count_stride;
    uint32_t dense;       // slots stored densely: maxv, or a multiple of kPageSlots below it
    uint32_t pool_pages;  // pages in the pool
    uint64_t pool_off;    // index in v of the pool's first double
    uint64_t cap;         // subpaths the page table is laid out for (stride of a level)
    uint32_t* page_tab;   // [level][cap]: page of subpath i for slots dense + level * kPageSlots ..
    uint32_t* pool_next;  // pages handed out in this batch; zeroed before every batch

    DEV uint64_t rec(uint32_t slot, uint64_t i) const {
        if (!PAGED || slot < dense) return ((uint64_t)slot * n + i) * kVertexDoubles;
        const uint32_t d = slot - dense;
        const uint32_t page = page_tab[(uint64_t)(d / kPageSlots) * cap + i];
        return pool_off + ((uint64_t)page * kPageSlots + d % kPageSlots) * kVertexDoubles;
    }
    // Before slot `slot` of subpath i is written for the first time: takes a page from the pool when the slot opens
    // one. False: the pool is empty.
    DEV bool openSlot(uint32_t slot, uint64_t i) const {
        if (slot < dense) return true;
        const uint32_t d = slot - dense;
        if (d % kPageSlots != 0) return true;
        const uint32_t page = atomicAdd(pool_next, 1u);
        if (page >= pool_pages) return false;
        page_tab[(uint64_t)(d / kPageSlots) * cap + i] = page;
        return true;
    }
    // pointer to the record of a slot: callers that read several fields of one vertex look the record up once
    DEV double* rp(uint32_t slot, uint64_t i) const { return v + rec(slot, i); }
    DEV uint64_t at(int f, uint32_t slot, uint64_t i) const { return rec(slot, i) + (uint64_t)f; }
    // object index and flags (bits 0..7 = lobe + 1, bit 8 = specular) share one 8-byte field
    DEV int32_t* oflag(uint32_t slot, uint64_t i) const { return (int32_t*)(v + rec(slot, i) + F_OBJFLAGS); }
    DEV int32_t obj(uint32_t slot, uint64_t i) const { return oflag(slot, i)[0]; }
    DEV int32_t flg(uint32_t slot, uint64_t i) const { return oflag(slot, i)[1]; }
    DEV void setObjFlags(uint32_t slot, uint64_t i, int32_t o, int32_t f) const { int32_t* p = oflag(slot, i); p[0] = o; p[1] = f; }
    DEV double ld(int f, uint32_t slot, uint64_t i) const { return v[at(f, slot, i)]; }
    DEV void st(int f, uint32_t slot, uint64_t i, double x) const { v[at(f, slot, i)] = x; }
    DEV d3 ld3(int f, uint32_t slot, uint64_t i) const {
        return mk3(v[at(f, slot, i)], v[at(f + 1, slot, i)], v[at(f + 2, slot, i)]);
    }
    DEV void st3(int f, uint32_t slot, uint64_t i, d3 x) const {
        v[at(f, slot, i)] = x.x; v[at(f + 1, slot, i)] = x.y; v[at(f + 2, slot, i)] = x.z;
    }
    DEV bool specular(uint32_t slot, uint64_t i) const { return (flg(slot, i) >> 8) & 1; }
    // the count words may be interleaved with other per-subpath state (count_stride words apart): the
    // wavefront kernels keep them inside the carry records, which they have to touch anyway
    DEV uint32_t meta(uint64_t i) const { return count[i * count_stride]; }
    DEV void setMeta(uint64_t i, uint32_t m) const { count[i * count_stride] = m; }
    DEV uint32_t cnt(uint64_t i) const { return meta(i) & 0xffu; }
    // may vertex k of subpath i (count word `meta`) be connected, i.e. is it not specular?
    DEV bool connectable(uint32_t meta, uint32_t k, uint64_t i) const {
        return k < kMaskedVertices ? !((meta >> (8 + k)) & 1u) : !specular(k, i);
    }
    DEV int lobe(uint32_t slot, uint64_t i) const { return (flg(slot, i) & 0xff) - 1; }
};
using PathStore = PathStoreT<true>;
using DensePathStore = PathStoreT<false>;
static_assert(sizeof(PathStore) == sizeof(DensePathStore), "the dense view has the fields of the store");
DEV DensePathStore denseView(const PathStore& P) {
    DensePathStore d;
    d.v = P.v; d.count = P.count; d.n = P.n; d.maxv = P.maxv; d.count_stride = P.count_stride; d.dense = P.dense;
    d.pool_pages = P.pool_pages; d.pool_off = P.pool_off; d.cap = P.cap; d.page_tab = P.page_tab; d.pool_next = P.pool_next;
    return d;
}

// fields of a record found through PathStore::rp
DEV d3 rld3(const double* r, int f) { return mk3(r[f], r[f + 1], r[f + 2]); }
DEV int32_t robj(const double* r) { return reinterpret_cast<const int32_t*>(r + F_OBJFLAGS)[0]; }
DEV int32_t rflg(const double* r) { return reinterpret_cast<const int32_t*>(r + F_OBJFLAGS)[1]; }
DEV bool rspecular(const double* r) { return (rflg(r) >> 8) & 1; }

// A vertex held in registers while it is being built.
struct Vertex {
    SurfacePoint sp;
    d3 a;
    d3 tex;  // textureSample at (sp.tex_u, sp.tex_v); unused for lens and emitter vertices
    bool specular;
    double p_light_A, p_eye_A;
};

// A specular vertex is never the end of a connection (ubdptrenderer.cpp:604): all that is ever read
// of it are the fields of line 0 (position for the neighbours' directions, the two pdfs and the flag
// for the MIS walks), so only that line is written.
// Records are written with 16-byte stores. (Marking these stores non-temporal, so that the streamed records would leave
// the tree in the caches, was measured on a 2 M-triangle scene: the ray tracing kernels stay where they are and the
// kernels that store slow down, generation from 3.4 to 8.6 ms per batch; plain stores stay.)
typedef double v2d __attribute__((ext_vector_type(2)));
DEV void st2(double* p, double a, double b) {
    v2d v; v.x = a; v.y = b;
    *reinterpret_cast<v2d*>(p) = v;
}
template <class ST>
DEV void storeVertex(const ST& P, uint32_t slot, uint64_t i, const Vertex& x) {
    double* r = P.rp(slot, i);
    double of;
    {
        int32_t w[2] = {x.sp.object, ((x.sp.lobe + 1) & 0xff) | (x.specular ? 0x100 : 0)};
        __builtin_memcpy(&of, w, 8);
    }
    st2(r + 0, x.sp.pos.x, x.sp.pos.y); st2(r + 2, x.sp.pos.z, x.p_light_A); st2(r + 4, x.p_eye_A, of);
    st2(r + 6, x.tex.x, x.tex.y);
    if (x.specular) return;
    st2(r + 8, x.tex.z, x.sp.Ng.x); st2(r + 10, x.sp.Ng.y, x.sp.Ng.z);
    st2(r + 12, x.sp.Ns.x, x.sp.Ns.y); st2(r + 14, x.sp.Ns.z, x.sp.Ts.x);
    st2(r + 16, x.sp.Ts.y, x.sp.Ts.z); st2(r + 18, x.sp.Bs.x, x.sp.Bs.y);
    st2(r + 20, x.sp.Bs.z, x.a.x); st2(r + 22, x.a.y, x.a.z);
}

// The few fields of the last stored vertex that the subpath loop keeps in registers.
struct Tail {
    d3 pos, Ns;
    double p_eye_A;
    bool specular;
};

// s0sample (uengine/ubdptrenderer.cpp:156-222). `n` = subpath.size(), `cur` = subpath.back(),
// `prev_specular` = subpath[n-2].specular; earlier vertices are read from the store.
template <class ST>
DEV d3 s0sample(const DScene& S, const ST& E, uint64_t i, uint32_t n, const Tail& cur, bool prev_specular,
                const Vertex& ev) {
    if (ev.sp.object < 0) return mk3(0, 0, 0);
    const DObject& e = S.objects[ev.sp.object];
    if (!e.is_emitter) return mk3(0, 0, 0);
    double w = 1.0;
    double ratio = (e.prob / e.area) / ev.p_eye_A;
    if (!cur.specular) w += ratio * ratio;
    d3 edge = cur.pos - ev.sp.pos;
    double edge_length2 = dot(edge, edge);
    edge = edge / sqrt(edge_length2);
    double d1 = dot(ev.sp.Ns, edge);
    double d2 = dot(cur.Ns, -edge);
    double G = fabs(d1 * d2) / edge_length2;
    double p_light_psa = 1.0 / (2.0 * kPi * fabs(d1));
    ratio *= (p_light_psa * G) / cur.p_eye_A;
    if (!cur.specular) {
        if (n > 1) {
            if (!prev_specular) w += ratio * ratio;
        } else
            w += ratio * ratio;
    }
    if (n > 2)
        for (int v = (int)n - 2; v > 0; v--) {
            const double* r = E.rp(v, i);
            ratio *= r[F_PL] / r[F_PE];
            if (rspecular(r)) continue;
            if (E.specular(v - 1, i)) continue;
            w += ratio * ratio;
        }
    w = 1.0 / w;
    d3 c = (e.power / e.area) * p_light_psa;
    return w * c * ev.a;
}

// ubdptrenderer.cpp:303-320 / 493-510: frame flip for transmission, LOCAL-space offset, to world.
DEV void finishVertex(const DObject& o, Vertex& nv, d3 next_dirT) {
    if (next_dirT.y < 0) {
        nv.sp.Ng = nv.sp.Ng * -1.0; nv.sp.Ns = nv.sp.Ns * -1.0;
        nv.sp.Ts = nv.sp.Ts * -1.0; nv.sp.Bs = nv.sp.Bs * -1.0;
    }
    nv.sp.pos = nv.sp.pos + 0.00001 * nv.sp.Ng;
    nv.sp.pos = transformPoint(o.W, nv.sp.pos);
    nv.sp.Ng = transformVectorT(o.invW, nv.sp.Ng);
    nv.sp.Ns = transformVectorT(o.invW, nv.sp.Ns);
    nv.sp.Ts = transformVectorT(o.invW, nv.sp.Ts);
    nv.sp.Bs = transformVectorT(o.invW, nv.sp.Bs);
}

// |d1 * d2| / len^2 with d1 = dot(edge, n_from), d2 = dot(-edge, n_to), edge = unit(to - from)
// (the G_prev / G_next blocks of ubdptrenderer.cpp:379-395, 568-584)
DEV double geomAbs(d3 from, d3 n_from, d3 to, d3 n_to) {
    d3 edge = to - from;
    double l2 = dot(edge, edge);
    edge = edge / sqrt(l2);
    double d1 = dot(edge, n_from);
    double d2 = dot(-edge, n_to);
    return fabs(d1 * d2) / l2;
}

// A stored vertex as the connection code needs it.
struct ConnVertex {
    d3 pos, Ng, Ns, Ts, Bs, a, tex;
    double p_light_A, p_eye_A;
    int32_t object;
    int lobe;
    bool specular;
};
template <class ST>
DEV void loadConn(const ST& P, uint32_t slot, uint64_t i, ConnVertex& c) {
    const double* r = P.rp(slot, i);
    c.pos = rld3(r, F_POS); c.Ng = rld3(r, F_NG); c.Ns = rld3(r, F_NS);
    c.Ts = rld3(r, F_TS);   c.Bs = rld3(r, F_BS); c.a = rld3(r, F_A);
    c.p_light_A = r[F_PL]; c.p_eye_A = r[F_PE];
    c.tex = rld3(r, F_TEX);
    c.object = robj(r);
    int f = rflg(r);
    c.lobe = (f & 0xff) - 1;
    c.specular = (f >> 8) & 1;
}
// The lens vertex (slot 0 of an eye subpath): only its position varies, so only line 0 of its record is stored
// and the rest is made here (the values computeEyeSubpath gives it, ubdptrenderer.cpp:231-246).
DEV d3 lensNormal(const DScene& S) { return transformVector(S.invV, mk3(0, 0, 1)); }
template <class ST>
DEV void loadConnEye(const DScene& S, const ST& E, uint32_t slot, uint64_t i, ConnVertex& c) {
    if (slot != 0) { loadConn(E, slot, i, c); return; }
    const double* r = E.rp(0, i);
    c.pos = rld3(r, F_POS);
    c.p_light_A = r[F_PL]; c.p_eye_A = r[F_PE];
    c.Ng = c.Ns = lensNormal(S);
    c.Ts = transformVector(S.invV, mk3(1, 0, 0));
    c.Bs = transformVector(S.invV, mk3(0, 1, 0));
    c.a = mk3(1.0, 1.0, 1.0);
    c.tex = mk3(0, 0, 0);
    c.object = -1; c.lobe = -1; c.specular = false;
}
template <class ST>
DEV void storeLensVertex(const ST& E, uint64_t i, const Vertex& x) {
    double* r = E.rp(0, i);
    r[F_POS] = x.sp.pos.x; r[F_POS + 1] = x.sp.pos.y; r[F_POS + 2] = x.sp.pos.z;
    r[F_PL] = x.p_light_A; r[F_PE] = x.p_eye_A;
    int32_t* of = reinterpret_cast<int32_t*>(r + F_OBJFLAGS);
    of[0] = x.sp.object; of[1] = ((x.sp.lobe + 1) & 0xff);
    r[F_TEX] = 0.0; r[F_TEX + 1] = 0.0;
}

DEV Frame frameOf(const ConnVertex& c) { Frame f; f.Ng = c.Ng; f.Ns = c.Ns; f.Ts = c.Ts; f.Bs = c.Bs; return f; }

// Projects a light vertex onto the image plane (ubdptrenderer.cpp:111-131). Returns false when the
// projection misses the image; otherwise the pixel the sample contributes to.
DEV bool t1Pixel(const DScene& S, d3 light_pos, d3 lens_pos, uint32_t& px, uint32_t& py) {
    d3 rayW = light_pos - lens_pos;
    d3 rayV = normalize(mul_xyz(S.V, rayW, 0.0));
    double d = S.image_plane_distance / rayV.z;
    d3 lens_posV = mul_xyz(S.V, lens_pos, 1.0);
    d3 ipV = lens_posV + d * rayV;
    double pu = 0.5 * ((ipV.x / S.image_plane_ratio) + 1);
    double pv = 1.0 - 0.5 * (ipV.y + 1);
    if (!((pu >= 0) && (pu <= 1) && (pv >= 0) && (pv <= 1))) return false;
    px = (uint32_t)floor((double)(S.img_res_x - 1) * pu);  // the reference maps with (W-1), (H-1)
    py = (uint32_t)floor((double)(S.img_res_y - 1) * pv);
    return true;
}

// The two same-side tests of UBsdfLambertian::samplePSA / pPSA (ubsdflambertian.cpp:17-23, 35-41) for
// the world-space directions a and b leaving vertex v. The reference normalises the tangent-space
// and the re-expanded vectors first; normalising scales by a positive number, so the signs tested
// here are the same.
DEV bool lambertSameSide(const ConnVertex& v, d3 a, d3 b) {
    m3 TNB = mk_m3(v.Ts, v.Ns, v.Bs);
    d3 aT = rowmul(a, TNB), bT = rowmul(b, TNB);
    if (aT.y * bT.y <= 0) return false;
    d3 aL = mul(TNB, aT), bL = mul(TNB, bT);
    return !(dot(v.Ng, aL) * dot(v.Ng, bL) <= 0);
}

// The first kWalkAhead steps of the two MIS walks of a connection depend on stored pdfs only (step
// k of the eye walk is vertex t-2-k, of the light walk vertex s-2-k). They are loaded WITHOUT
// branches (a step that does not exist reads vertex 0 instead and is never used), so that the
// caller can issue them together with the loads of the two end vertices: one round trip to memory
// instead of one per step after the connection's arithmetic. Vertex 0 of either subpath (lens,
// emitter) is never specular.
#ifndef UBDPT_WALK_AHEAD
#define UBDPT_WALK_AHEAD 3
#endif
constexpr int kWalkAhead = UBDPT_WALK_AHEAD;
struct WalkAhead {
    double plE[kWalkAhead], peE[kWalkAhead], plL[kWalkAhead], peL[kWalkAhead];
    uint32_t specE, specL;  // bit k: the vertex of step k is specular
};
template <class ST>
DEV void loadWalkAhead(const ST& L, const ST& E, uint64_t i, uint32_t s, uint32_t t, WalkAhead& a) {
    a.specE = a.specL = 0;
#pragma unroll
    for (int k = 0; k < kWalkAhead; k++) {
        int v = (int)t - 2 - k;
        uint32_t vc = v > 0 ? (uint32_t)v : 0u;
        const double* r = E.rp(vc, i);
        a.plE[k] = r[F_PL];
        a.peE[k] = r[F_PE];
        a.specE |= ((v > 0) & ((rflg(r) >> 8) & 1)) ? (1u << k) : 0u;
        v = (int)s - 2 - k;
        vc = v > 0 ? (uint32_t)v : 0u;
        r = L.rp(vc, i);
        a.plL[k] = r[F_PL];
        a.peL[k] = r[F_PE];
        a.specL |= ((v > 0) & ((rflg(r) >> 8) & 1)) ? (1u << k) : 0u;
    }
}

// Lambertian-only form of connectionFactorTail + misWeight below: a connected vertex is never
// specular (ubdptrenderer.cpp:604), and every non-specular surface vertex carries the Lambertian
// lobe, whose value (tex/pi) and pdf (1/pi) are constants once the same-side tests pass.
// Returns false when the connection contributes nothing; otherwise c (connection factor) and w (MIS weight).
template <class ST>
DEV bool connectLambert(const DScene& S, const ST& L, const ST& E, uint64_t i, const ConnVertex& vl,
                        const ConnVertex& ve, uint32_t s, uint32_t t, d3 lprev, d3 eprev, const WalkAhead& wa, d3& c,
                        double& w) {
    const uint32_t specE = wa.specE, specL = wa.specL;
    d3 ce = vl.pos - ve.pos;
    double ce_length2 = dot(ce, ce);
    ce = ce / sqrt(ce_length2);
    double d1 = dot(ce, ve.Ng);
    double d2 = dot(-ce, vl.Ng);
    if ((d1 <= 0) || (d2 <= 0)) return false;
    d1 = dot(ce, ve.Ns);
    d2 = dot(-ce, vl.Ns);
    double G = (d1 * d2) / ce_length2;
    d3 fs1, fs2;
    if (s == 1)
        fs1 = mk3(1, 1, 1) * (1.0 / (2.0 * kPi * d2));
    else {
        if (!lambertSameSide(vl, lprev - vl.pos, ve.pos - vl.pos)) return false;
        fs1 = (1.0 / kPi) * vl.tex;
        if (fs1.x + fs1.y + fs1.z <= 0) return false;
    }
    if (t == 1) {
        d3 edgeV = mul_xyz(S.V, ce, 0.0);
        double d = S.image_plane_distance / edgeV.z;
        double d2_image_plane = dot(-edgeV, mk3(0, 0, -1));
        double G_image_plane = fabs(d1 * d2_image_plane) / (d * d);
        fs2 = mk3(1, 1, 1) * ((1.0 / (S.image_plane_area)) / G_image_plane);
    } else {
        if (!lambertSameSide(ve, vl.pos - ve.pos, eprev - ve.pos)) return false;
        fs2 = (1.0 / kPi) * ve.tex;
        if (fs2.x + fs2.y + fs2.z <= 0) return false;
    }
    c = fs1 * fs2 * G;
    // weight (ubdptrenderer.cpp:779-844); the pdfs of p_sp1 / p_sm1 are 1/pi because the same-side
    // tests above are exactly the tests of pPSA for the same pair of directions
    w = 1.0;
    double ratio = 1.0;
    if (t > 1) {
        ratio = (s == 1) ? (1.0 / (2.0 * kPi * d2)) : (1.0 / kPi);
        ratio *= G;
        ratio = ratio / ve.p_eye_A;
        if (!(specE & 1u)) w += ratio * ratio;
    }
    if (t > 2) {
        int v = (int)t - 2;
#pragma unroll
        for (int k = 0; k < kWalkAhead; k++, v--) {
            if (v < 1) break;
            ratio *= wa.plE[k] / wa.peE[k];
            if ((specE >> k) & 1u) continue;
            if (k + 1 < kWalkAhead ? ((specE >> (k + 1)) & 1u) != 0 : (v - 1 > 0 && E.specular(v - 1, i))) continue;
            w += ratio * ratio;
        }
        for (; v > 0; v--) {
            const double* r = E.rp(v, i);
            ratio *= r[F_PL] / r[F_PE];
            if (rspecular(r)) continue;
            if (v - 1 > 0 && E.specular(v - 1, i)) continue;
            w += ratio * ratio;
        }
    }
    double p_eye_A = (t == 1) ? (1.0 / S.image_plane_area) : ((1.0 / kPi) * G);
    ratio = (p_eye_A / vl.p_light_A);
    if (s > 1) {
        if (!(specL & 1u)) w += ratio * ratio;
    } else
        w += ratio * ratio;
    if (s > 1) {
        int v = (int)s - 2;
#pragma unroll
        for (int k = 0; k < kWalkAhead; k++, v--) {
            if (v < 0) break;
            ratio *= wa.peL[k] / wa.plL[k];
            if ((specL >> k) & 1u) continue;
            if (v > 0)
                if (k + 1 < kWalkAhead ? ((specL >> (k + 1)) & 1u) != 0 : (v - 1 > 0 && L.specular(v - 1, i))) continue;
            w += ratio * ratio;
        }
        for (; v >= 0; v--) {
            const double* r = L.rp(v, i);
            ratio *= r[F_PE] / r[F_PL];
            if (rspecular(r)) continue;
            if (v > 1)
                if (L.specular(v - 1, i)) continue;
            w += ratio * ratio;
        }
    }
    w = 1.0 / w;
    return true;
}

// connectionFactor after the specular and visibility tests (ubdptrenderer.cpp:612-690), any lobe.
// lprev = light_subpath[s-2].pos (s > 1), eprev = eye_subpath[t-2].pos (t > 1).
DEV bool connectionFactorTail(const DScene& S, const ConnVertex& vl, const ConnVertex& ve, uint32_t s, uint32_t t,
                              d3 lprev, d3 eprev, d3& c) {
    d3 ce = vl.pos - ve.pos;
    double ce_length2 = dot(ce, ce);
    ce = ce / sqrt(ce_length2);
    double d1 = dot(ce, ve.Ng);
    double d2 = dot(-ce, vl.Ng);
    if ((d1 <= 0) || (d2 <= 0)) return false;
    d1 = dot(ce, ve.Ns);
    d2 = dot(-ce, vl.Ns);
    double G = (d1 * d2) / ce_length2;
    d3 fs1, fs2;
    if (s == 1)
        fs1 = mk3(1, 1, 1) * (1.0 / (2.0 * kPi * d2));
    else {
        m3 TNB = mk_m3(vl.Ts, vl.Ns, vl.Bs);
        fs1 = bsdfEval(vl.lobe, S.objects[vl.object].mat_a, vl.tex, frameOf(vl), normalize(rowmul(lprev - vl.pos, TNB)), normalize(rowmul(ve.pos - vl.pos, TNB)));
        if (fs1.x + fs1.y + fs1.z <= 0) return false;
    }
    if (t == 1) {
        d3 edgeV = mul_xyz(S.V, ce, 0.0);
        double d = S.image_plane_distance / edgeV.z;
        double d2_image_plane = dot(-edgeV, mk3(0, 0, -1));
        double G_image_plane = fabs(d1 * d2_image_plane) / (d * d);
        fs2 = mk3(1, 1, 1) * ((1.0 / (S.image_plane_area)) / G_image_plane);
    } else {
        m3 TNB = mk_m3(ve.Ts, ve.Ns, ve.Bs);
        fs2 = bsdfEval(ve.lobe, S.objects[ve.object].mat_a, ve.tex, frameOf(ve), normalize(rowmul(vl.pos - ve.pos, TNB)), normalize(rowmul(eprev - ve.pos, TNB)));
        if (fs2.x + fs2.y + fs2.z <= 0) return false;
    }
    c = fs1 * fs2 * G;
    return true;
}

// weight (ubdptrenderer.cpp:779-844) with p_sp1 (:693-735) and p_sm1 (:737-777) inlined.
template <class SL, class SE>
DEV double misWeight(const DScene& S, const SL& L, const SE& E, uint64_t i, const ConnVertex& vl,
                     const ConnVertex& ve, uint32_t s, uint32_t t, d3 lprev, d3 eprev) {
    double w = 1.0;
    double ratio = 1.0;
    // shared geometry of p_sp1 / p_sm1
    d3 ce = vl.pos - ve.pos;
    double ce_length2 = dot(ce, ce);
    ce = ce / sqrt(ce_length2);
    double d1 = dot(ce, ve.Ns);
    double d2 = dot(-ce, vl.Ns);
    double G = (d1 * d2) / ce_length2;
    if (t > 1) {
        // p_sp1
        if (s == 1)
            ratio = (1.0 / (2.0 * kPi * d2));
        else {
            m3 TNB = mk_m3(vl.Ts, vl.Ns, vl.Bs);
            d3 wsT = rowmul(normalize(ve.pos - vl.pos), TNB);
            d3 wgT = rowmul(normalize(lprev - vl.pos), TNB);
            ratio = bsdfPdf(vl.lobe, S.objects[vl.object].mat_a, frameOf(vl), wsT, wgT);
        }
        ratio *= G;
        ratio = ratio / ve.p_eye_A;
        if (!E.specular(t - 2, i)) w += ratio * ratio;
    }
    if (t > 2)
        for (int v = (int)t - 2; v > 0; v--) {
            const double* r = E.rp(v, i);
            ratio *= r[F_PL] / r[F_PE];
            if (rspecular(r)) continue;
            if (E.specular(v - 1, i)) continue;
            w += ratio * ratio;
        }
    // p_sm1
    double p_eye_A;
    if (t == 1)
        p_eye_A = (1.0 / S.image_plane_area);
    else {
        m3 TNB = mk_m3(ve.Ts, ve.Ns, ve.Bs);
        d3 wsT = rowmul(normalize(vl.pos - ve.pos), TNB);
        d3 wgT = rowmul(normalize(eprev - ve.pos), TNB);
        p_eye_A = (bsdfPdf(ve.lobe, S.objects[ve.object].mat_a, frameOf(ve), wsT, wgT) * G);
    }
    ratio = (p_eye_A / vl.p_light_A);
    if (s > 1) {
        if (!L.specular(s - 2, i)) w += ratio * ratio;
    } else
        w += ratio * ratio;
    if (s > 1)
        for (int v = (int)s - 2; v >= 0; v--) {
            const double* r = L.rp(v, i);
            ratio *= r[F_PE] / r[F_PL];
            if (rspecular(r)) continue;
            if (v > 0)
                if (L.specular(v - 1, i)) continue;
            w += ratio * ratio;
        }
    return (1.0 / w);
}

}  // namespace ub
