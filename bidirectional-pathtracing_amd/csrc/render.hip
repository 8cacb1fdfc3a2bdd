// render.hip — the BDPT pass on the device (gfx950).
//
// One pixel-sample = one UBDPTRenderer::renderPixel call (uengine/ubdptrenderer.cpp:88-154): an eye
// subpath, a light subpath, all (s,t) connections with power-heuristic MIS, the s=0 emission hits
// and the t=1 light-tracer splats. A batch of `n_passes` passes runs concurrently (pass index is
// part of the path id), results are added to a planar fp64 sum buffer with hardware f64 atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>

#include "paths.h"
#include "render.h"

namespace ub {

constexpr int kBlock = 256;


struct Tally {
    uint32_t closest = 0, shadow = 0, conn = 0, splat = 0, capped = 0, fb = 0;
    TraceCounters tc{0, 0, 0, 0, 0};
};

DEV Stack makeStack(int32_t* lds_base, int32_t* ovf, uint64_t ovf_stride, uint64_t gid, uint32_t stack_cap) {
    Stack st;
    st.lds = lds_base + threadIdx.x;
    st.lds_stride = kBlock;
    st.ovf = ovf + gid;
    st.ovf_stride = ovf_stride;
    st.sp = 0;
    st.cap = (int)(stack_cap < (uint32_t)(kLdsStack + kOverflowStack) ? stack_cap : (uint32_t)(kLdsStack + kOverflowStack));
    st.overflowed = false;
    return st;
}

// EXACT: the kernel has staged the full-precision tree in LDS (lds_scene.h) and walks that one
template <bool STATS, bool EXACT = false>
DEV Closest closestHit(const DScene& S, const Ray& r, Stack& st, Tally& ty) {
    if (STATS) ty.closest++;
    if (S.accel == 1) return closestHitLinear(S, r, STATS ? &ty.tc : nullptr);
    return closestHitBvh<EXACT>(S, r, st, STATS ? &ty.tc : nullptr);
}
template <bool STATS, bool EXACT = false>
DEV bool visible(const DScene& S, d3 p0, d3 p1, Stack& st, Tally& ty) {
    if (STATS) ty.shadow++;
    if (S.accel == 1) return visibleLinear(S, p0, p1, STATS ? &ty.tc : nullptr);
    return visibleBvh<EXACT>(S, p0, p1, st, STATS ? &ty.tc : nullptr);
}

DEV void flushTally(const Tally& ty, unsigned long long* counters, uint32_t eye_v, uint32_t light_v) {
    atomicAdd(&counters[C_CLOSEST], (unsigned long long)ty.closest);
    atomicAdd(&counters[C_SHADOW], (unsigned long long)ty.shadow);
    atomicAdd(&counters[C_CONN], (unsigned long long)ty.conn);
    atomicAdd(&counters[C_SPLAT], (unsigned long long)ty.splat);
    atomicAdd(&counters[C_NODES], (unsigned long long)ty.tc.nodes);
    atomicAdd(&counters[C_PRIMS], (unsigned long long)ty.tc.prims);
    if (ty.tc.wave_nodes) atomicAdd(&counters[C_WAVE_NODES], (unsigned long long)ty.tc.wave_nodes);
    if (ty.tc.wave_prims) atomicAdd(&counters[C_WAVE_PRIMS], (unsigned long long)ty.tc.wave_prims);
    if (ty.tc.max_stack) atomicMax(&counters[C_MAX_STACK], (unsigned long long)ty.tc.max_stack);
    atomicAdd(&counters[C_CAPPED], (unsigned long long)ty.capped);
    atomicAdd(&counters[C_FB], (unsigned long long)ty.fb);
    atomicAdd(&counters[C_EYE_V], (unsigned long long)eye_v);
    atomicAdd(&counters[C_LIGHT_V], (unsigned long long)light_v);
}

// hardware f64 atomic add into the planar sum buffer
DEV void fbAdd(double* fb, uint64_t npix, uint64_t pixel, d3 v) {
    unsafeAtomicAdd(fb + pixel, v.x);
    unsafeAtomicAdd(fb + npix + pixel, v.y);
    unsafeAtomicAdd(fb + 2 * npix + pixel, v.z);
}

// Pixel-sample gid of the batch -> (index of its pass among the call's passes, pixel). The batch starts at pixel
// batch_pixel0 of pass index batch_pass0; batch_pixel0 + gid and the pixel count almost always fit 32 bits, where
// a division costs a fraction of the 64-bit one (every stage of every bounce decodes its path for the generator key).
DEV void decodePassPixel(const RenderArgs& A, uint64_t gid, uint64_t& pass_index, uint64_t& pixel) {
    const uint64_t npix = (uint64_t)A.S.img_res_x * A.S.img_res_y;
    const uint64_t t = A.batch_pixel0 + gid;
    uint64_t q;
    if (((t | npix) >> 32) == 0) {
        const uint32_t q32 = (uint32_t)t / (uint32_t)npix;
        q = q32;
        pixel = (uint32_t)t - q32 * (uint32_t)npix;
    } else {
        q = t / npix;
        pixel = t - q * npix;
    }
    pass_index = A.batch_pass0 + q;
}

// With fb_slots > 0 the passes of a call are kept apart: the q-th pass the batch touches adds into the lane's sum buffer
// q (3 * npix doubles each). Returns the offset for the pixel-sample gid of the batch; it is added to the pixel index
// handed to fbAdd.
DEV uint64_t fbSlotOffset(const RenderArgs& A, uint64_t gid) {
    if (A.fb_slots == 0) return 0;
    uint64_t pass_index, pixel;
    decodePassPixel(A, gid, pass_index, pixel);
    return (pass_index - A.batch_pass0) * 3ull * ((uint64_t)A.S.img_res_x * A.S.img_res_y);
}

// A subpath found the pool of deep vertex slots empty and ends here: the batch cannot be completed as it is.
DEV void poolExhausted(const RenderArgs& A) {
    atomicAdd(&A.counters[C_POOL_OVERFLOW], 1ULL);
    *A.batch_fail = 1u;
}

DEV void decodePath(const RenderArgs& A, uint64_t gid, uint64_t& pass, uint64_t& pixel, uint32_t& px, uint32_t& py) {
    // a batch is a contiguous range of the call's pixel-samples in (pass, pixel) order; it need not start or
    // end on a pass boundary
    uint64_t pi;
    decodePassPixel(A, gid, pi, pixel);
    pass = A.first_pass + pi * A.pass_stride;
    // pixel < npix <= 2^32 and the width is below 2^16: a 32-bit division
    py = (uint32_t)pixel / A.S.img_res_x;
    px = (uint32_t)pixel - py * A.S.img_res_x;
}

// Everything between "a hit was found" and "the next ray is ready" is identical for both subpaths
// (ubdptrenderer.cpp:285-336 and 474-523).
struct Bounce {
    d3 fs, dirT;
    double p_psa;
};

// ---------------------------------------------------------------------------------------------------
// eye subpath: computeEyeSubpath (ubdptrenderer.cpp:224-407), primary ray = :231-257
// ---------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_eye(RenderArgs A) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= A.E.n) return;
    const DScene& S = A.S;
    const PathStore& E = A.E;
    uint64_t pass, pixel;
    uint32_t px, py;
    decodePath(A, gid, pass, pixel, px, py);
    Rng rng{pathKey(A.seed, pass, pixel), STREAM_EYE};
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    uint64_t npix = (uint64_t)S.img_res_x * S.img_res_y;

    uint32_t pixel_sample = (uint32_t)(pass % S.num_pixel_strata);
    uint32_t lens_sample = (uint32_t)(pass % S.num_lens_strata);

    d3 I = mk3(0, 0, 0);
    d2 disk = sampleUnitDiskStratified(S.num_lens_strata, lens_sample, rng(DIM_LENS_X), rng(DIM_LENS_Y));
    d3 lens_pointV = mk3(S.lens_radius * disk.x, S.lens_radius * disk.y, 0);

    Vertex lens;
    lens.a = mk3(1.0, 1.0, 1.0);  // m_W
    lens.tex = mk3(0, 0, 0);
    lens.sp.pos = mul_xyz(S.invV, lens_pointV, 1.0);
    lens.sp.Ns = lens.sp.Ng = transformVector(S.invV, mk3(0, 0, 1));
    lens.sp.Ts = transformVector(S.invV, mk3(1, 0, 0));
    lens.sp.Bs = transformVector(S.invV, mk3(0, 1, 0));
    lens.sp.tex_u = 0; lens.sp.tex_v = 0; lens.sp.object = -1; lens.sp.lobe = LOBE_NONE; lens.sp.face = 0;
    lens.p_eye_A = 1.0 / S.lens_area;  // lens_area is 0 in the reference (never assigned) -> +inf
    lens.p_light_A = 0;
    lens.specular = false;
    storeVertex(E, 0, gid, lens);
    uint32_t n = 1, smask = 0;

    d2 pp = sampleUnitRectStratified(S.num_pixel_strata, pixel_sample, rng(DIM_PIXEL_X), rng(DIM_PIXEL_Y));
    d3 image_pointV = mk3(-S.image_plane_ratio + ((double)px + pp.x) * S.pixel_width,
                          1.0 - ((double)py + pp.y) * S.pixel_height, S.image_plane_distance);
    image_pointV = normalize(image_pointV);
    d3 focus_plane_pointV = image_pointV * (S.focus_plane_distance / image_pointV.z);
    Ray ray;
    ray.o = lens.sp.pos;
    ray.d = transformVector(S.invV, focus_plane_pointV - lens_pointV);

    Tail cur;   // subpath.back()
    cur.pos = lens.sp.pos; cur.Ns = lens.sp.Ns; cur.p_eye_A = lens.p_eye_A; cur.specular = false;
    d3 prev_pos = mk3(0, 0, 0), prev_Ns = mk3(0, 0, 0);  // subpath[n-2]
    bool prev_specular = false;

    uint32_t h = 1;
    Vertex nv;  // next_vertex
    nv.specular = false; nv.p_light_A = 0; nv.p_eye_A = 0; nv.a = mk3(0, 0, 0);
    Closest hit = closestHit<STATS>(S, ray, st, ty);
    bool alive = hit.object >= 0;
    if (alive) {
        surfacePoint(S, ray, hit, nv.sp);
        nv.sp.lobe = pickLobe(S.objects[nv.sp.object], rng(hitDim(h, DIM_LOBE)));
        nv.a = lens.a;
        nv.p_eye_A = 1.0 / S.image_plane_area;
    }
    while (alive) {
        const DObject& obj = S.objects[nv.sp.object];
        if (nv.sp.lobe == LOBE_NONE) {  // :273-283
            nv.sp.pos = transformPoint(obj.W, nv.sp.pos);
            nv.sp.Ns = transformVectorT(obj.invW, nv.sp.Ns);
            I = I + s0sample(S, E, gid, n, cur, prev_specular, nv);
            break;
        }
        if (n >= S.max_path_vertices) { if (STATS) ty.capped++; break; }
        d3 lTs = nv.sp.Ts, lNs = nv.sp.Ns, lBs = nv.sp.Bs;  // local, unflipped frame of the float TNB (:287-290)
        Frame fr; fr.Ng = nv.sp.Ng; fr.Ns = nv.sp.Ns; fr.Ts = nv.sp.Ts; fr.Bs = nv.sp.Bs;
        d3 w = transformVector(obj.invW, -ray.d);
        d3 tex = textureSample(S, obj.texture, nv.sp.tex_u, nv.sp.tex_v);
        nv.tex = tex;
        double p_psa;
        d3 fs, dirT;
        if (!bsdfScatter(nv.sp.lobe, obj.mat_a, tex, fr, w, rng(hitDim(h, DIM_SCATTER_1)), rng(hitDim(h, DIM_SCATTER_2)),
                         dirT, p_psa, fs, nv.specular))
            break;
        finishVertex(obj, nv, dirT);
        I = I + s0sample(S, E, gid, n, cur, prev_specular, nv);
        if (!E.openSlot(n, gid)) { poolExhausted(A); break; }
        storeVertex(E, n, gid, nv);
        smask |= specularBit(n, nv.specular);
        n++;
        // the pushed vertex becomes curr, the old curr becomes prev
        prev_pos = cur.pos; prev_Ns = cur.Ns; prev_specular = cur.specular;
        cur.pos = nv.sp.pos; cur.Ns = nv.sp.Ns; cur.p_eye_A = nv.p_eye_A; cur.specular = nv.specular;
        d3 cur_a = nv.a;

        ray.o = cur.pos;
        ray.d = transformVector(obj.W, float_tnb_mul(lTs, lNs, lBs, dirT));
        uint32_t h_curr = h;
        h++;
        hit = closestHit<STATS>(S, ray, st, ty);
        if (hit.object < 0) break;
        surfacePoint(S, ray, hit, nv.sp);
        nv.sp.lobe = pickLobe(S.objects[nv.sp.object], rng(hitDim(h, DIM_LOBE)));

        double fs_sum = fs.x + fs.y + fs.z;
        if (!(p_psa > 0) || !(fs_sum > 0)) break;
        double q;
        if (n < S.min_depth) q = 1.0;
        else {
            q = fmin(1.0, (fs_sum / 3.0) / p_psa);
            if (rng(hitDim(h_curr, DIM_ROULETTE)) > q) break;
        }
        if (cur.specular) {
            nv.a = cur_a * fs;
            nv.p_eye_A = q * p_psa;
            E.st(F_PL, n - 2, gid, q * p_psa);
        } else {
            d3 a = fs / p_psa;
            double G_prev = geomAbs(cur.pos, cur.Ns, prev_pos, prev_Ns);
            // the reference uses the new hit's OBJECT-LOCAL position and normal here (:389-393)
            double G_next = geomAbs(cur.pos, cur.Ns, nv.sp.pos, nv.sp.Ns);
            nv.a = cur_a * a;
            nv.p_eye_A = q * p_psa * G_next;
            E.st(F_PL, n - 2, gid, q * p_psa * G_prev);
        }
    }
    E.setMeta(gid, n | smask);
    if (I.x != 0 || I.y != 0 || I.z != 0) { fbAdd(A.fb, npix, pixel + fbSlotOffset(A, gid), I); if (STATS) ty.fb++; }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// ---------------------------------------------------------------------------------------------------
// light subpath: computeLightSubpath (ubdptrenderer.cpp:409-594); emitter point = uapp/emitter.cpp:19-30
// ---------------------------------------------------------------------------------------------------
constexpr uint64_t kLinearCdfFaces = 64;  // emitter meshes up to this size are sampled with the reference's linear walk
DEV void emitterRandomPoint(const DScene& S, const DObject& e, const Rng& rng, d3& pos, d3& Ng, d3& Ns, d3& Ts, d3& Bs) {
    if (e.kind == 0) {  // uapp/implicitsphere.cpp:32-38
        Ns = Ng = sampleUnitSphereUniform(rng(DIM_POINT_A), rng(DIM_POINT_B));
        pos = Ng;
        Ts = -Ns + mk3(0, 0, 1.0 / Ns.z);
        Bs = cross(Ns, Ts);
    } else {  // uapp/mesh.cpp:206-232: walk over the face probabilities until the uniform number is used up
        double r = rng(DIM_POINT_A);
        uint64_t f = 0;
        if (e.num_faces <= kLinearCdfFaces) {  // the reference's own walk, subtraction by subtraction
            for (; f + 1 < e.num_faces; f++) {
                double p = S.face_prob[e.first_face + f];
                if (r < p) break;
                r -= p;
            }
        } else {
            // Large meshes: the face whose range of the running sum holds r, by bisection (the walk is O(#faces) per
            // light subpath). The walk subtracts face by face and the table adds face by face, so within a few
            // hundred ulps of a boundary between two faces the two can pick different neighbours: a fraction of
            // about 1e-13 * #faces of the samples, each moving to an adjacent face of the same mesh.
            const double* cdf = S.face_cdf + e.first_face;
            uint64_t lo = 0, hi = e.num_faces - 1;  // the answer is in [lo, hi]; the last face takes what rounding leaves over
            while (lo < hi) {
                const uint64_t mid = (lo + hi + 1) >> 1;
                if (cdf[mid] <= r) lo = mid; else hi = mid - 1;
            }
            f = lo;
        }
        const float* p = S.face_pos + 9 * (e.first_face + f);
        const FaceAttr& fa = S.face_attr[e.first_face + f];
        d3 p0 = loadP(p), p1 = loadP(p + 3), p2 = loadP(p + 6);
        double u, v;
        pos = sampleTriangleUniform(p0, p1, p2, rng(DIM_POINT_B), rng(DIM_POINT_C), u, v);
        Ns = normalize((1.0 - u - v) * loadP(fa.n) + u * loadP(fa.n + 3) + v * loadP(fa.n + 6));
        Ng = normalize(cross(p1 - p0, p2 - p0));
        if (dot(Ns, Ng) < 0) Ng = Ng * -1.0;
        Ts = normalize((1.0 - u - v) * loadP(fa.t) + u * loadP(fa.t + 3) + v * loadP(fa.t + 6));
        Bs = normalize(cross(Ns, Ts));
    }
    pos = pos + Ng * 0.0001;
    pos = transformPoint(e.W, pos);
    Ng = transformVectorT(e.invW, Ng);
    Ns = transformVectorT(e.invW, Ns);
    Ts = transformVectorT(e.invW, Ts);
    Bs = transformVectorT(e.invW, Bs);
}

template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_light(RenderArgs A) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= A.L.n) return;
    const DScene& S = A.S;
    const PathStore& L = A.L;
    uint64_t pass, pixel;
    uint32_t px, py;
    decodePath(A, gid, pass, pixel, px, py);
    Rng rng{pathKey(A.seed, pass, pixel), STREAM_LIGHT};
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    uint32_t n = 0, smask = 0;

    if (S.num_emitters > 0) {
        // emitter choice by cumulative probability (:413-428)
        double p = rng(DIM_EMITTER);
        int32_t ei = -1;
        for (uint32_t k = 0; k < S.num_emitters; k++) {
            int32_t e = S.emitters[k];
            double pe = S.objects[e].prob;
            if (p < pe) { ei = e; break; }
            p -= pe;
        }
        if (ei < 0) ei = S.emitters[S.num_emitters - 1];
        const DObject& em = S.objects[ei];

        Vertex ev;
        emitterRandomPoint(S, em, rng, ev.sp.pos, ev.sp.Ng, ev.sp.Ns, ev.sp.Ts, ev.sp.Bs);
        ev.sp.tex_u = 0; ev.sp.tex_v = 0; ev.sp.object = ei; ev.sp.lobe = LOBE_NONE; ev.sp.face = 0;
        ev.a = em.power;
        ev.tex = mk3(0, 0, 0);
        ev.p_light_A = em.prob * (1.0 / em.area);
        ev.p_eye_A = 0;  // indeterminate in the reference (:433), zero here
        ev.specular = false;
        storeVertex(L, 0, gid, ev);
        n = 1;

        d3 dirT = samplePosHemUniform(rng(DIM_EMIT_ANGLE), rng(DIM_EMIT_V));
        Ray ray;
        ray.o = ev.sp.pos;
        ray.d = float_tnb_mul(ev.sp.Ts, ev.sp.Ns, ev.sp.Bs, dirT);  // float TNB (:448-453), not normalised

        Tail cur;
        cur.pos = ev.sp.pos; cur.Ns = ev.sp.Ns; cur.p_eye_A = 0; cur.specular = false;
        d3 prev_pos = mk3(0, 0, 0), prev_Ns = mk3(0, 0, 0);

        uint32_t h = 1;
        Vertex nv;
        nv.specular = false; nv.p_light_A = 0; nv.p_eye_A = 0; nv.a = mk3(0, 0, 0);
        Closest hit = closestHit<STATS>(S, ray, st, ty);
        bool alive = hit.object >= 0;
        if (alive) {
            surfacePoint(S, ray, hit, nv.sp);
            nv.sp.lobe = pickLobe(S.objects[nv.sp.object], rng(hitDim(h, DIM_LOBE)));
            alive = nv.sp.lobe != LOBE_NONE;
        }
        if (alive) {
            // G with the hit still in object-local space (:463-472)
            double G = geomAbs(ev.sp.pos, ev.sp.Ns, nv.sp.pos, nv.sp.Ns);
            nv.a = ev.a;
            nv.p_light_A = (1 / (2.0 * kPi * dirT.y)) * G;
        }
        while (alive) {
            if (n >= S.max_path_vertices) { if (STATS) ty.capped++; break; }
            const DObject& obj = S.objects[nv.sp.object];
            d3 lTs = nv.sp.Ts, lNs = nv.sp.Ns, lBs = nv.sp.Bs;
            Frame fr; fr.Ng = nv.sp.Ng; fr.Ns = nv.sp.Ns; fr.Ts = nv.sp.Ts; fr.Bs = nv.sp.Bs;
            d3 w = transformVector(obj.invW, -ray.d);
            d3 tex = textureSample(S, obj.texture, nv.sp.tex_u, nv.sp.tex_v);
            nv.tex = tex;
            double p_psa;
            d3 fs, dT;
            if (!bsdfScatter(nv.sp.lobe, obj.mat_a, tex, fr, w, rng(hitDim(h, DIM_SCATTER_1)),
                             rng(hitDim(h, DIM_SCATTER_2)), dT, p_psa, fs, nv.specular))
                break;
            finishVertex(obj, nv, dT);
            if (!L.openSlot(n, gid)) { poolExhausted(A); break; }
            storeVertex(L, n, gid, nv);
            smask |= specularBit(n, nv.specular);
            n++;
            prev_pos = cur.pos; prev_Ns = cur.Ns;
            cur.pos = nv.sp.pos; cur.Ns = nv.sp.Ns; cur.specular = nv.specular;
            d3 cur_a = nv.a;

            ray.o = cur.pos;
            ray.d = transformVector(obj.W, float_tnb_mul(lTs, lNs, lBs, dT));
            uint32_t h_curr = h;
            h++;
            hit = closestHit<STATS>(S, ray, st, ty);
            if (hit.object < 0) break;
            surfacePoint(S, ray, hit, nv.sp);
            nv.sp.lobe = pickLobe(S.objects[nv.sp.object], rng(hitDim(h, DIM_LOBE)));
            if (nv.sp.lobe == LOBE_NONE) break;

            double fs_sum = fs.x + fs.y + fs.z;
            if (!(p_psa > 0) || !(fs_sum > 0)) break;
            double q;
            if (n < S.min_depth) q = 1.0;
            else {
                q = fmin(1.0, (fs_sum / 3.0) / p_psa);
                if (rng(hitDim(h_curr, DIM_ROULETTE)) > q) break;
            }
            if (cur.specular) {
                nv.a = cur_a * fs;
                nv.p_light_A = q * 1.0;  // the light path stores q, the eye path q * p_psa (:557-558 vs :366-367)
                L.st(F_PE, n - 2, gid, q * 1.0);
            } else {
                d3 a = fs / p_psa;
                double G_prev = geomAbs(prev_pos, prev_Ns, cur.pos, cur.Ns);
                double G_next = geomAbs(cur.pos, cur.Ns, nv.sp.pos, nv.sp.Ns);  // local-space next vertex (:578-582)
                nv.a = cur_a * a;
                nv.p_light_A = q * p_psa * G_next;
                L.st(F_PE, n - 2, gid, q * p_psa * G_prev);
            }
        }
    }
    L.setMeta(gid, n | smask);
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

}  // namespace ub
#include "lds_scene.h"
#include "wavefront.h"
namespace ub {

// ---------------------------------------------------------------------------------------------------
// connections: the (s,t) loop of renderPixel (ubdptrenderer.cpp:103-149), one pixel-sample per lane
// ---------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_connect(RenderArgs A) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= A.E.n) return;
    const DScene& S = A.S;
    const PathStore& E = A.E;
    const PathStore& L = A.L;
    uint64_t pass, pixel;
    uint32_t px, py;
    decodePath(A, gid, pass, pixel, px, py);
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    uint64_t npix = (uint64_t)S.img_res_x * S.img_res_y;
    uint32_t nE = E.cnt(gid), nL = L.cnt(gid);
    if (STATS) { Tally t0; flushTally(t0, A.counters, nE, nL); }
    d3 I = mk3(0, 0, 0);
    d3 lens_pos = E.ld3(F_POS, 0, gid);
    d3 lens_a = mk3(1.0, 1.0, 1.0);  // the lens vertex's throughput (only line 0 of its record is stored)
    for (uint32_t s = 1; s <= nL; s++) {
        ConnVertex vl;
        loadConn(L, s - 1, gid, vl);
        if (vl.specular) continue;  // every t is rejected (:604)
        d3 lprev = s > 1 ? L.ld3(F_POS, s - 2, gid) : mk3(0, 0, 0);
        for (uint32_t t = 1; t <= nE; t++) {
            uint32_t tx = 0, tyy = 0;
            if (t == 1) {
                if (!t1Pixel(S, vl.pos, lens_pos, tx, tyy)) continue;
            }
            if (E.specular(t - 1, gid)) continue;
            ConnVertex ve;
            loadConnEye(S, E, t - 1, gid, ve);
            if (!visible<STATS>(S, vl.pos, ve.pos, st, ty)) continue;
            d3 eprev = t > 1 ? E.ld3(F_POS, t - 2, gid) : mk3(0, 0, 0);
            d3 c;
            if (!connectionFactorTail(S, vl, ve, s, t, lprev, eprev, c)) continue;
            double w = misWeight(S, L, E, gid, vl, ve, s, t, lprev, eprev);
            if (STATS) ty.conn++;
            if (t == 1) {
                d3 I_t1 = vl.a * lens_a * c * w;
                if (STATS) { ty.splat++; ty.fb++; }
                fbAdd(A.fb, npix, (uint64_t)tyy * S.img_res_x + tx + fbSlotOffset(A, gid), I_t1);
            } else
                I = I + vl.a * ve.a * c * w;
        }
    }
    if (I.x != 0 || I.y != 0 || I.z != 0) { fbAdd(A.fb, npix, pixel + fbSlotOffset(A, gid), I); if (STATS) ty.fb++; }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// ---------------------------------------------------------------------------------------------------
// connections, wavefront form: the (s,t) pairs of all pixel-samples of the batch are flattened into
// one queue (wave-level prefix sum + one atomic per wave to reserve queue space), then one lane
// evaluates one pair. Lanes no longer wait for the longest (s,t) loop in their wave.
// ---------------------------------------------------------------------------------------------------
DEV uint64_t packPair(uint64_t i, uint32_t s, uint32_t t) { return (i << 16) | ((uint64_t)s << 8) | t; }

// inclusive prefix sum across the 64 lanes of a wave
DEV uint32_t waveInclusiveScan(uint32_t v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Connection factor c and MIS weight w of the pair (s, t) of pixel-sample i, after the tests that reject without
// arithmetic (specular ends, t = 1 projection, shadow ray): connectionFactor from its cosine tests on
// (ubdptrenderer.cpp:612-690) and weight (:779-844). Every load whose address depends on (i, s, t) only is issued
// before the first use of any of them (the positions before the end vertices exist only for s, t > 1 and are only
// used then; slot 0 is read otherwise). a = throughput product of the two end vertices. False: contributes nothing.
template <class ST>
DEV bool pairConnection(const DScene& S, const ST& L, const ST& E, uint64_t i, uint32_t s, uint32_t t,
                        ConnVertex& vl, ConnVertex& ve, d3& c, double& w) {
    loadConn(L, s - 1, i, vl);
    loadConnEye(S, E, t - 1, i, ve);
    d3 lprev = L.ld3(F_POS, s > 1 ? s - 2 : 0, i);
    d3 eprev = E.ld3(F_POS, t > 1 ? t - 2 : 0, i);
    WalkAhead wa;
    loadWalkAhead(L, E, i, s, t, wa);
    if ((s == 1 || vl.lobe == LOBE_LAMBERTIAN) && (t == 1 || ve.lobe == LOBE_LAMBERTIAN))
        return connectLambert(S, L, E, i, vl, ve, s, t, lprev, eprev, wa, c, w);
    if (!connectionFactorTail(S, vl, ve, s, t, lprev, eprev, c)) return false;
    w = misWeight(S, L, E, i, vl, ve, s, t, lprev, eprev);
    return true;
}

// Second half of a connection (no ray is traced here). Returns the contribution and the pixel it belongs to;
// false when the pair contributes nothing.
// DEEP = false: s and t lie within the dense vertex slots (then so does every vertex the pair touches).
template <bool STATS, bool DEEP>
DEV bool pairEval(const RenderArgs& A, uint64_t i, uint32_t s, uint32_t t, Tally& ty, uint64_t& pixel, d3& value) {
    const DScene& S = A.S;
    ConnVertex vl, ve;
    d3 c;
    double w;
    bool ok;
    if (DEEP) ok = pairConnection(S, A.L, A.E, i, s, t, vl, ve, c, w);
    else      ok = pairConnection(S, denseView(A.L), denseView(A.E), i, s, t, vl, ve, c, w);
    uint32_t tx = 0, tyy = 0;
    if (t == 1) {
        if (!t1Pixel(S, vl.pos, ve.pos, tx, tyy)) return false;
    }
    if (!ok) return false;
    if (STATS) { ty.conn++; ty.fb++; }
    if (t == 1) {
        if (STATS) ty.splat++;
        pixel = (uint64_t)tyy * S.img_res_x + tx;
    } else {
        uint64_t pass_index;
        decodePassPixel(A, i, pass_index, pixel);
    }
    pixel += fbSlotOffset(A, i);  // index into the sum buffers of all slots
    value = vl.a * ve.a * c * w;
    return true;
}

template <bool STATS>
DEV void connectPair(const RenderArgs& A, uint64_t i, uint32_t s, uint32_t t, Stack& st, Tally& ty) {
    const DScene& S = A.S;
    const PathStore& E = A.E;
    const PathStore& L = A.L;
    uint64_t npix = (uint64_t)S.img_res_x * S.img_res_y;
    d3 lpos = L.ld3(F_POS, s - 1, i);
    d3 epos = E.ld3(F_POS, t - 1, i);
    uint32_t tx = 0, tyy = 0;
    if (t == 1) {
        if (!t1Pixel(S, lpos, epos, tx, tyy)) return;
    }
    {
        d3 ce = lpos - epos;
        ce = ce / sqrt(dot(ce, ce));
        if ((dot(ce, t == 1 ? lensNormal(S) : E.ld3(F_NG, t - 1, i)) <= 0) || (dot(-ce, L.ld3(F_NG, s - 1, i)) <= 0)) return;
    }
    if (!visible<STATS>(S, lpos, epos, st, ty)) return;
    ConnVertex vl, ve;
    loadConn(L, s - 1, i, vl);
    loadConnEye(S, E, t - 1, i, ve);
    d3 lprev = s > 1 ? L.ld3(F_POS, s - 2, i) : mk3(0, 0, 0);
    d3 eprev = t > 1 ? E.ld3(F_POS, t - 2, i) : mk3(0, 0, 0);
    WalkAhead wa;
    loadWalkAhead(L, E, i, s, t, wa);
    d3 c;
    double w;
    if ((s == 1 || vl.lobe == LOBE_LAMBERTIAN) && (t == 1 || ve.lobe == LOBE_LAMBERTIAN)) {
        if (!connectLambert(S, L, E, i, vl, ve, s, t, lprev, eprev, wa, c, w)) return;
    } else {
        if (!connectionFactorTail(S, vl, ve, s, t, lprev, eprev, c)) return;
        w = misWeight(S, L, E, i, vl, ve, s, t, lprev, eprev);
    }
    if (STATS) { ty.conn++; ty.fb++; }
    if (t == 1) {
        if (STATS) ty.splat++;
        fbAdd(A.fb, npix, (uint64_t)tyy * S.img_res_x + tx + fbSlotOffset(A, i), vl.a * ve.a * c * w);
    } else {
        uint64_t pass_index, pixel;
        decodePassPixel(A, i, pass_index, pixel);
        fbAdd(A.fb, npix, pixel + fbSlotOffset(A, i), vl.a * ve.a * c * w);
    }
}

// One lane per pixel-sample: appends its connectable (s,t) pairs to the pair queue. The specular
// bits of the first kMaskedVertices vertices travel in the subpath's count word, so the common case
// reads two words per pixel-sample (from the dense copies the subpath kernels write when a subpath ends:
// consecutive lanes read consecutive words); queue space is reserved with one atomic per workgroup.
// A pixel-sample whose pairs no longer fit the queue goes to a list instead (A.wf_trace_q, free once the subpaths
// are complete; its length is A.ray_count[2]) and k_pair_overflow evaluates its pairs one by one: same result,
// and this kernel stays small (the slow path inside it cost it three quarters of its occupancy).
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_pair_enum(RenderArgs A) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x;
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + tid;
    bool active = gid < A.E.n;
    const PathStore& E = A.E;
    const PathStore& L = A.L;
    uint32_t mE = 0, mL = 0, nE = 0, nL = 0, cE = 0, cL = 0;
    if (active) {
        mE = A.fin_E[gid]; mL = A.fin_L[gid];
        nE = mE & 0xff; nL = mL & 0xff;
        for (uint32_t t = 0; t < nE; t++) cE += E.connectable(mE, t, gid) ? 1u : 0u;
        for (uint32_t s = 0; s < nL; s++) cL += L.connectable(mL, s, gid) ? 1u : 0u;
    }
    uint32_t cnt = cE * cL;
    if (STATS && active) { Tally t0; flushTally(t0, A.counters, nE, nL); }
    uint32_t incl = waveInclusiveScan(cnt);
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) tot += s_wave[w];
        s_base = tot ? atomicAdd(A.pair_count, tot) : 0ULL;
    }
    __syncthreads();
    unsigned long long wave_base = s_base;
    for (uint32_t w = 0; w < (tid >> 6); w++) wave_base += s_wave[w];
    // (Listing a wave's pairs in LDS first and writing its part of the queue with consecutive lanes on
    // consecutive entries was measured: same kernel time, so every lane writes its own short run.)
    if (!cnt) return;
    unsigned long long pos = wave_base + (incl - cnt);
    if (pos + cnt <= A.pair_capacity) {
        // t-major: the t = 1 pairs (splats into other pixels) first, then every pair that lands in the
        // pixel-sample's own pixel next to each other, so that k_pair_eval can add them up before its atomics
        for (uint32_t t = 1; t <= nE; t++) {
            if (!E.connectable(mE, t - 1, gid)) continue;
            for (uint32_t s = 1; s <= nL; s++) {
                if (!L.connectable(mL, s - 1, gid)) continue;
                A.pairs[pos++] = packPair(gid, s, t);
            }
        }
    } else {
        // queue full: the part of this lane's reservation that lies inside the queue is filled with skip records
        // (s == 0), and the pixel-sample is listed for k_pair_overflow
        for (unsigned long long j = pos; j < pos + cnt && j < A.pair_capacity; j++) A.pairs[j] = 0;
        A.wf_trace_q[atomicAdd(&A.ray_count[2], 1u)] = (uint32_t)gid;
    }
}

// What the queues could not take, after everything else of the batch: the pixel-samples k_pair_enum had no room for
// (every lane runs the connection loop of its pixel-sample, shadow rays included, the reference's loop structure) and
// the visible pairs with an end vertex beyond the dense vertex slots, which k_pair_eval put aside. Launched once per
// batch; both lists are almost always empty.
// (Compiled for four waves per SIMD instead, 128 registers with 471 spilled, so that a workgroup of it would start wherever
// one workgroup of the other lane's ray tracing kernels ends: 305 against 307 Mpaths/s on a 2 M-triangle scene, no gain.)
#ifndef UBDPT_RARE_WAVES
#define UBDPT_RARE_WAVES 1
#endif
template <bool STATS>
__global__ __launch_bounds__(kBlock, UBDPT_RARE_WAVES) void k_pair_overflow(RenderArgs A) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint64_t gid0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint32_t total = A.ray_count[2];
    const uint32_t deep_total = A.ray_count[3];
    if (total == 0 && deep_total == 0) return;
    const PathStore& E = A.E;
    const PathStore& L = A.L;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid0, A.ovf_cap);
    Tally ty;
    // the visible pairs k_pair_eval put aside because an end vertex lies beyond the dense vertex slots
    const uint64_t npix_all = (uint64_t)A.S.img_res_x * A.S.img_res_y;
    for (uint64_t j = gid0; j < deep_total; j += nthreads) {
        const uint64_t p = A.ray_pairs[j];
        uint64_t pixel;
        d3 v;
        if (pairEval<STATS, true>(A, p >> 16, (uint32_t)((p >> 8) & 0xff), (uint32_t)(p & 0xff), ty, pixel, v)) fbAdd(A.fb, npix_all, pixel, v);
    }
    for (uint64_t j = gid0; j < total; j += nthreads) {
        const uint64_t gid = A.wf_trace_q[j];
        const uint32_t mE = E.meta(gid), mL = L.meta(gid);
        const uint32_t nE = mE & 0xff, nL = mL & 0xff;
        for (uint32_t s = 1; s <= nL; s++) {
            if (!L.connectable(mL, s - 1, gid)) continue;
            for (uint32_t t = 1; t <= nE; t++) {
                if (!E.connectable(mE, t - 1, gid)) continue;
                connectPair<STATS>(A, gid, s, t, st, ty);
            }
        }
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

}  // namespace ub
#include "wavefront_split.h"
namespace ub {

// ---------------------------------------------------------------------------------------------------
// probes: batch intersection / visibility / BSDF evaluation for parity checks
// ---------------------------------------------------------------------------------------------------
// Grid-stride: lane `gid` of the launch owns slice `gid` of the overflow stack and takes rays gid, gid + lanes, ...
// (the ordinary entry point launches one lane per ray, the entry point that allocates nothing a fixed number of lanes).
__global__ __launch_bounds__(kBlock) void k_intersect_batch(DScene S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits,
                                                           int32_t* ovf, uint64_t ovf_stride, uint32_t ovf_cap,
                                                           unsigned long long* counters) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t lanes = (uint64_t)gridDim.x * kBlock;
    Stack st = makeStack(lds_stack, ovf, ovf_stride, gid, ovf_cap);
    Tally ty;
    for (uint64_t i = gid; i < n; i += lanes) {
        Ray r;
        r.o = mk3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = mk3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        Closest c = closestHit<true>(S, r, st, ty);
        ubdpt_hit h;
        if (c.object >= 0) {
            SurfacePoint sp;
            surfacePoint(S, r, c, sp);
            h.hit = 1; h.object = c.object; h.face = c.face == kSphereFace ? -1 : (int64_t)c.face; h.d = c.d;
            h.pos[0] = sp.pos.x; h.pos[1] = sp.pos.y; h.pos[2] = sp.pos.z;
            h.Ng[0] = sp.Ng.x; h.Ng[1] = sp.Ng.y; h.Ng[2] = sp.Ng.z;
            h.Ns[0] = sp.Ns.x; h.Ns[1] = sp.Ns.y; h.Ns[2] = sp.Ns.z;
            h.Ts[0] = sp.Ts.x; h.Ts[1] = sp.Ts.y; h.Ts[2] = sp.Ts.z;
            h.Bs[0] = sp.Bs.x; h.Bs[1] = sp.Bs.y; h.Bs[2] = sp.Bs.z;
            h.tex_u = sp.tex_u; h.tex_v = sp.tex_v;
        } else {
            h.hit = 0; h.object = -1; h.face = -1; h.d = __builtin_inf();
            for (int k = 0; k < 3; k++) h.pos[k] = h.Ng[k] = h.Ns[k] = h.Ts[k] = h.Bs[k] = 0;
            h.tex_u = h.tex_v = 0;
        }
        hits[i] = h;
    }
    if (ty.tc.nodes) atomicAdd(&counters[C_NODES], (unsigned long long)ty.tc.nodes);
    if (ty.tc.prims) atomicAdd(&counters[C_PRIMS], (unsigned long long)ty.tc.prims);
    if (ty.tc.max_stack) atomicMax(&counters[C_MAX_STACK], (unsigned long long)ty.tc.max_stack);
    if (st.overflowed) atomicAdd(&counters[C_STACK_OVERFLOW], 1ULL);
}

__global__ __launch_bounds__(kBlock) void k_visibility_batch(DScene S, const ubdpt_segment* segs, size_t n,
                                                            uint8_t* vis, int32_t* ovf, uint64_t ovf_stride, uint32_t ovf_cap,
                                                            unsigned long long* counters) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t lanes = (uint64_t)gridDim.x * kBlock;
    Stack st = makeStack(lds_stack, ovf, ovf_stride, gid, ovf_cap);
    Tally ty;
    for (uint64_t i = gid; i < n; i += lanes) {
        d3 p0 = mk3(segs[i].p0[0], segs[i].p0[1], segs[i].p0[2]);
        d3 p1 = mk3(segs[i].p1[0], segs[i].p1[1], segs[i].p1[2]);
        vis[i] = visible<true>(S, p0, p1, st, ty) ? 1 : 0;
    }
    if (ty.tc.nodes) atomicAdd(&counters[C_NODES], (unsigned long long)ty.tc.nodes);
    if (ty.tc.prims) atomicAdd(&counters[C_PRIMS], (unsigned long long)ty.tc.prims);
    if (st.overflowed) atomicAdd(&counters[C_STACK_OVERFLOW], 1ULL);
}

// ---- connection probe: the connection code of the rendering kernels on one pair of subpaths handed in as arrays ----
// Fills the records of a one-path store from UPathVertex-shaped input (texture colour looked up from the
// vertex's texture coordinates, as shading does when it stores a vertex).
__global__ void k_probe_fill(DScene S, const ubdpt_path_vertex* in, uint32_t n, PathStore P) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) {
        uint32_t m = n;
        for (uint32_t j = 0; j < n && j < kMaskedVertices; j++) m |= specularBit(j, in[j].specular != 0);
        P.setMeta(0, m);
    }
    if (k >= n) return;
    const ubdpt_path_vertex& v = in[k];
    Vertex x;
    x.sp.pos = mk3(v.pos[0], v.pos[1], v.pos[2]); x.sp.Ng = mk3(v.Ng[0], v.Ng[1], v.Ng[2]);
    x.sp.Ns = mk3(v.Ns[0], v.Ns[1], v.Ns[2]); x.sp.Ts = mk3(v.Ts[0], v.Ts[1], v.Ts[2]); x.sp.Bs = mk3(v.Bs[0], v.Bs[1], v.Bs[2]);
    x.sp.tex_u = v.tex_u; x.sp.tex_v = v.tex_v; x.sp.object = v.object; x.sp.lobe = v.lobe; x.sp.face = 0;
    x.a = mk3(v.a[0], v.a[1], v.a[2]);
    x.tex = v.object >= 0 ? textureSample(S, S.objects[v.object].texture, v.tex_u, v.tex_v) : mk3(0, 0, 0);
    x.specular = false;  // store the whole record; the flag follows
    x.p_light_A = v.p_light_A; x.p_eye_A = v.p_eye_A;
    storeVertex(P, k, 0, x);
    P.setObjFlags(k, 0, v.object, ((v.lobe + 1) & 0xff) | (v.specular ? 0x100 : 0));
}

__global__ __launch_bounds__(kBlock) void k_connection_probe(DScene S, PathStore E, PathStore L, uint32_t n_eye, uint32_t n_light,
                                                            ubdpt_connection* out, double* s0, int32_t* ovf, uint64_t ovf_stride,
                                                            uint32_t ovf_cap, unsigned long long* counters) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint64_t lane0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t lanes = (uint64_t)gridDim.x * kBlock;
    Stack st = makeStack(lds_stack, ovf, ovf_stride, lane0, ovf_cap);
    Tally ty;
    const uint32_t pairs = n_eye * n_light;
    const uint64_t work = (uint64_t)pairs + n_eye;
    for (uint64_t gid = lane0; gid < work; gid += lanes) {
    if (gid < pairs) {
        const uint32_t s = (uint32_t)gid / n_eye + 1, t = (uint32_t)gid % n_eye + 1;
        ubdpt_connection o;
        o.connected = 0; o.reserved = 0; o.c[0] = o.c[1] = o.c[2] = 0; o.w = 0;
        const uint32_t mE = E.meta(0), mL = L.meta(0);
        // connectionFactor's order: specular ends (:604), visibility (:609), then the rest
        if (L.connectable(mL, s - 1, 0) && E.connectable(mE, t - 1, 0) &&
            visible<true>(S, L.ld3(F_POS, s - 1, 0), E.ld3(F_POS, t - 1, 0), st, ty)) {
            ConnVertex vl, ve;
            d3 c;
            double w;
            if (pairConnection(S, L, E, 0, s, t, vl, ve, c, w)) {
                o.connected = 1; o.c[0] = c.x; o.c[1] = c.y; o.c[2] = c.z; o.w = w;
            }
        }
        out[gid] = o;
    } else if (s0) {
        const uint32_t t = (uint32_t)(gid - pairs);
        d3 v = mk3(0, 0, 0);
        if (t >= 1) {
            ConnVertex cv;
            loadConn(E, t, 0, cv);
            Vertex ev;
            ev.sp.object = cv.object; ev.sp.pos = cv.pos; ev.sp.Ns = cv.Ns; ev.a = cv.a; ev.p_eye_A = cv.p_eye_A;
            Tail cur;
            cur.pos = E.ld3(F_POS, t - 1, 0);
            cur.Ns = t - 1 == 0 ? lensNormal(S) : E.ld3(F_NS, t - 1, 0);
            cur.p_eye_A = E.ld(F_PE, t - 1, 0);
            cur.specular = E.specular(t - 1, 0);
            const bool prev_specular = t >= 2 && E.specular(t - 2, 0);
            v = s0sample(S, E, 0, t, cur, prev_specular, ev);
        }
        s0[3 * t] = v.x; s0[3 * t + 1] = v.y; s0[3 * t + 2] = v.z;
    }
    }
    if (st.overflowed) atomicAdd(&counters[C_STACK_OVERFLOW], 1ULL);
}

__global__ __launch_bounds__(kBlock) void k_bsdf_batch(int mode, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* out) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    const ubdpt_bsdf_query& Q = q[gid];
    Frame fr;
    fr.Ng = mk3(Q.frame[0], Q.frame[1], Q.frame[2]); fr.Ns = mk3(Q.frame[3], Q.frame[4], Q.frame[5]);
    fr.Ts = mk3(Q.frame[6], Q.frame[7], Q.frame[8]); fr.Bs = mk3(Q.frame[9], Q.frame[10], Q.frame[11]);
    d3 tex = mk3(Q.tex[0], Q.tex[1], Q.tex[2]);
    d3 wi = mk3(Q.wi[0], Q.wi[1], Q.wi[2]), wo = mk3(Q.wo[0], Q.wo[1], Q.wo[2]);
    ubdpt_bsdf_result r;
    r.f[0] = r.f[1] = r.f[2] = 0; r.pdf = 0; r.dir[0] = r.dir[1] = r.dir[2] = 0; r.ok = 0; r.specular = 0;
    if (mode == 0) {
        d3 f = bsdfEval(Q.lobe, Q.eta, tex, fr, wi, wo);
        r.f[0] = f.x; r.f[1] = f.y; r.f[2] = f.z;
    } else if (mode == 1) {
        r.pdf = bsdfPdf(Q.lobe, Q.eta, fr, wi, wo);
    } else {
        d3 dir, fs;
        double pdf;
        bool spec = false;
        if (bsdfScatter(Q.lobe, Q.eta, tex, fr, wi, Q.u1, Q.u2, dir, pdf, fs, spec)) {
            r.ok = 1; r.specular = spec ? 1 : 0; r.pdf = pdf;
            r.f[0] = fs.x; r.f[1] = fs.y; r.f[2] = fs.z;
            r.dir[0] = dir.x; r.dir[1] = dir.y; r.dir[2] = dir.z;
        }
    }
    out[gid] = r;
}

// planar device sum buffer -> interleaved [y][x][rgb] (UBuffer2D<dvec3> layout, uengine/uutils.h:37-98)
__global__ void k_interleave(const double* fb, uint64_t npix, double* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    out[3 * i] = fb[i];
    out[3 * i + 1] = fb[npix + i];
    out[3 * i + 2] = fb[2 * npix + i];
}
// the same, added to what `out` already holds (per-GPU partial sums of contexts that share a device)
__global__ void k_interleave_add(const double* fb, uint64_t npix, double* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    out[3 * i] += fb[i];
    out[3 * i + 1] += fb[npix + i];
    out[3 * i + 2] += fb[2 * npix + i];
}
// End of a batch (see launchCommitBatch in render.h). The other lane's batches may be committing into the same buffers:
// atomic adds. Most values of a lane's buffers are non-zero after a batch; the zeros are skipped all the same.
__global__ __launch_bounds__(kBlock) void k_commit_batch(double* lane_fb, uint64_t n, double* target, const uint32_t* fail,
                                                         unsigned long long* lane_counters, unsigned long long* counters,
                                                         uint32_t counter_words) {
    const bool failed = *fail != 0;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    if (gid < counter_words) {
        const unsigned long long v = lane_counters[gid];
        if (v) {
            lane_counters[gid] = 0;
            const uint32_t which = (uint32_t)gid % kNumCounters;
            if (which == C_MAX_STACK) { if (!failed) atomicMax(&counters[gid], v); }
            else if (!failed || which == C_POOL_OVERFLOW) atomicAdd(&counters[gid], v);
        }
    }
    for (uint64_t i = gid; i < n; i += stride) {
        const double v = lane_fb[i];
        if (v != 0) {   // (true for NaN as well)
            lane_fb[i] = 0;
            if (!failed) unsafeAtomicAdd(target + i, v);
        }
    }
}
__global__ void k_deinterleave(const double* in, uint64_t npix, double* fb) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    fb[i] = in[3 * i];
    fb[npix + i] = in[3 * i + 1];
    fb[2 * npix + i] = in[3 * i + 2];
}

// UEngine::imageRGB (uengine/uengine.cpp:182-202) per pixel: radiance = sum / passes, CIE XYZ through the
// three-bin fit sums, XYZ -> linear sRGB, clamp, gamma, clamp (uengine/uconverter.cpp:78-126).
// Reads the planar sum buffer, writes [y][x][rgb].
__global__ void k_image_rgb(const double* fb, uint64_t npix, double passes, double gamma, RgbFit fit, double* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    d3 radiance = mk3(fb[i], fb[npix + i], fb[2 * npix + i]) / passes;
    d3 r123 = radiance / 123.0;
    d3 xyz = mk3(dot(r123, fit.sumX), dot(r123, fit.sumY), dot(r123, fit.sumZ));
    double c[3] = {dot(xyz, mk3(3.2404542, -1.5371385, -0.4985314)), dot(xyz, mk3(-0.9692660, 1.8760108, 0.0415560)),
                   dot(xyz, mk3(0.0556434, -0.2040259, 1.0572252))};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = fmin(fmax(c[k], 0.0), 1.0);
        if (v <= 0.0031308) v *= 12.92;
        else v = 1.055 * pow(v, 1.0 / gamma) - 0.055;
        out[3 * i + k] = fmin(fmax(v, 0.0), 1.0);
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline unsigned gridFor(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

void launchEye(const RenderArgs& A, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_eye<true>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_eye<false>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
}
void launchLight(const RenderArgs& A, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_light<true>, dim3(gridFor(A.L.n)), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_light<false>, dim3(gridFor(A.L.n)), dim3(kBlock), 0, s, A);
}
void launchConnect(const RenderArgs& A, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_connect<true>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_connect<false>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
}
void launchWfGen(const RenderArgs& A, bool stats, hipStream_t s) {
    unsigned g = gridFor(2 * A.E.n);
    if (stats) hipLaunchKernelGGL(k_wf_gen<true>, dim3(g), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_wf_gen<false>, dim3(g), dim3(kBlock), 0, s, A);
}
// picks the instantiation (counting or not, how much of the scene fits into LDS) of a kernel template K<bool, int>
#define UBDPT_LAUNCH_TRACE(K, ...)                                                                             \
    do {                                                                                                       \
        const int level_ = ldsLevelFor(A.S);                                                                   \
        if (stats) {                                                                                           \
            if (level_ == 2) hipLaunchKernelGGL((K<true, 2>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);  \
            else if (level_ == 1) hipLaunchKernelGGL((K<true, 1>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<true, 0>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);              \
        } else {                                                                                               \
            if (level_ == 2) hipLaunchKernelGGL((K<false, 2>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__); \
            else if (level_ == 1) hipLaunchKernelGGL((K<false, 1>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<false, 0>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);             \
        }                                                                                                      \
    } while (0)
void launchWfTail(const RenderArgs& A, uint32_t depth, unsigned blocks, bool stats, hipStream_t s) {
    UBDPT_LAUNCH_TRACE(k_wf_tail, A, depth);
}
// `blocks`: grid of the kernel that finishes the pixel-samples the queue had no room for (its lanes index the
// traversal stacks' overflow buffer, which the caller sized for that grid)
void launchPairEnum(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_pair_enum<true>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_pair_enum<false>, dim3(gridFor(A.E.n)), dim3(kBlock), 0, s, A);
}
// picks the instantiation (counting or not, object table in LDS or not) of a kernel template K<bool, uint32_t>
#define UBDPT_LAUNCH_LDS(K, CAP, ...)                                                                          \
    do {                                                                                                       \
        const bool lds_ = A.S.num_objects <= (CAP);                                                            \
        if (stats) {                                                                                           \
            if (lds_) hipLaunchKernelGGL((K<true, (CAP)>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);     \
            else hipLaunchKernelGGL((K<true, 0u>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);             \
        } else {                                                                                               \
            if (lds_) hipLaunchKernelGGL((K<false, (CAP)>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);    \
            else hipLaunchKernelGGL((K<false, 0u>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);            \
        }                                                                                                      \
    } while (0)
// the persistent kernels exist for scenes in HBM (placement 0 and 1); a scene that fits LDS takes the plain ones
#define UBDPT_LAUNCH_PERSISTENT(K, ...)                                                                        \
    do {                                                                                                       \
        const int level_ = ldsLevelFor(A.S);                                                                   \
        if (stats) {                                                                                           \
            if (level_ == 1) hipLaunchKernelGGL((K<true, 1>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);  \
            else hipLaunchKernelGGL((K<true, 0>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);              \
        } else {                                                                                               \
            if (level_ == 1) hipLaunchKernelGGL((K<false, 1>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<false, 0>), dim3(blocks), dim3(kBlock), 0, s, __VA_ARGS__);             \
        }                                                                                                      \
    } while (0)
void launchWfTrace(const RenderArgs& A, uint32_t depth, unsigned blocks, unsigned small_scene_blocks, bool stats, hipStream_t s) {
    if (ldsLevelFor(A.S) == 2) {
        if (stats) hipLaunchKernelGGL(k_wf_trace_lds<true>, dim3(small_scene_blocks), dim3(kBlock), 0, s, A, depth);
        else hipLaunchKernelGGL(k_wf_trace_lds<false>, dim3(small_scene_blocks), dim3(kBlock), 0, s, A, depth);
        return;
    }
    UBDPT_LAUNCH_PERSISTENT(k_wf_trace, A, depth);
}
void launchPairVisibleLds(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_pair_visible_lds<true>, dim3(blocks), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_pair_visible_lds<false>, dim3(blocks), dim3(kBlock), 0, s, A);
}
void launchWfShade(const RenderArgs& A, uint32_t depth, unsigned blocks, bool stats, hipStream_t s) {
    UBDPT_LAUNCH_LDS(k_wf_shade, kLdsObjectsShade, A, depth);
}
int sceneLdsLevel(const DScene& S) { return ldsLevelFor(S); }
void launchPairSetup(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_pair_setup<true>, dim3(blocks), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_pair_setup<false>, dim3(blocks), dim3(kBlock), 0, s, A);
}
void launchRaysVisible(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s) {
    UBDPT_LAUNCH_PERSISTENT(k_rays_visible, A);
}
void launchPairCompact(const RenderArgs& A, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_pair_compact, dim3(blocks), dim3(kBlock), 0, s, A);
}
void launchPairEval(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL(k_pair_eval<true>, dim3(blocks), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_pair_eval<false>, dim3(blocks), dim3(kBlock), 0, s, A);
    // what the queues could not take (k_pair_overflow): one workgroup per CU at most, the lists are almost always empty,
    // and a workgroup of that kernel needs all registers of a SIMD to start at all
    const unsigned oblocks = blocks < 256u ? blocks : 256u;
    if (stats) hipLaunchKernelGGL(k_pair_overflow<true>, dim3(oblocks), dim3(kBlock), 0, s, A);
    else hipLaunchKernelGGL(k_pair_overflow<false>, dim3(oblocks), dim3(kBlock), 0, s, A);
}
void launchIntersectBatch(const DScene& S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits, int32_t* ovf,
                          uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s) {
    if (n == 0) return;
    // grid-stride: never more lanes than the overflow buffer has slices (ovf_stride, a multiple of the workgroup size)
    const unsigned blocks = (unsigned)std::min<uint64_t>(gridFor(n), std::max<uint64_t>(1, ovf_stride / kBlock));
    hipLaunchKernelGGL(k_intersect_batch, dim3(blocks), dim3(kBlock), 0, s, S, rays, n, hits, ovf, ovf_stride, ovf_cap, counters);
}
void launchVisibilityBatch(const DScene& S, const ubdpt_segment* segs, size_t n, uint8_t* vis, int32_t* ovf,
                           uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s) {
    if (n == 0) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(gridFor(n), std::max<uint64_t>(1, ovf_stride / kBlock));
    hipLaunchKernelGGL(k_visibility_batch, dim3(blocks), dim3(kBlock), 0, s, S, segs, n, vis, ovf, ovf_stride, ovf_cap, counters);
}
void launchConnectionProbe(const DScene& S, const ubdpt_path_vertex* eye, uint32_t n_eye, const ubdpt_path_vertex* light,
                           uint32_t n_light, PathStore E, PathStore L, ubdpt_connection* out, double* s0, int32_t* ovf,
                           uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(k_probe_fill, dim3(1), dim3(kBlock), 0, s, S, eye, n_eye, E);
    hipLaunchKernelGGL(k_probe_fill, dim3(1), dim3(kBlock), 0, s, S, light, n_light, L);
    const uint64_t threads = (uint64_t)n_eye * n_light + n_eye;
    const unsigned blocks = (unsigned)std::min<uint64_t>(gridFor(threads), std::max<uint64_t>(1, ovf_stride / kBlock));
    hipLaunchKernelGGL(k_connection_probe, dim3(blocks), dim3(kBlock), 0, s, S, E, L, n_eye, n_light, out, s0, ovf,
                       ovf_stride, ovf_cap, counters);
}
void launchBsdfBatch(int mode, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_bsdf_batch, dim3(gridFor(n)), dim3(kBlock), 0, s, mode, q, n, r);
}
void launchInterleave(const double* fb, uint64_t npix, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_interleave, dim3(gridFor(npix)), dim3(kBlock), 0, s, fb, npix, out);
}
void launchCommitBatch(double* lane_fb, uint64_t npix, uint32_t slots, double* target, const uint32_t* fail,
                       unsigned long long* lane_counters, unsigned long long* counters, uint32_t counter_words, hipStream_t s) {
    const uint64_t n = 3ull * npix * slots;
    const unsigned blocks = (unsigned)std::min<uint64_t>(2048, std::max<uint64_t>((n + kBlock - 1) / kBlock, (counter_words + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_commit_batch, dim3(blocks), dim3(kBlock), 0, s, lane_fb, n, target, fail, lane_counters, counters, counter_words);
}
void launchInterleaveAdd(const double* fb, uint64_t npix, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_interleave_add, dim3(gridFor(npix)), dim3(kBlock), 0, s, fb, npix, out);
}
void launchDeinterleave(const double* in, uint64_t npix, double* fb, hipStream_t s) {
    hipLaunchKernelGGL(k_deinterleave, dim3(gridFor(npix)), dim3(kBlock), 0, s, in, npix, fb);
}
void launchImageRgb(const double* fb, uint64_t npix, double passes, double gamma, const RgbFit& fit, double* out,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_image_rgb, dim3(gridFor(npix)), dim3(kBlock), 0, s, fb, npix, passes, gamma, fit, out);
}


// ---------------------------------------------------------------------------------------------------
// Scratch memory of the entry point that allocates nothing (ubdpt_intersect_batch_noalloc): the overflow part of
// the traversal stacks of kNoAllocLanes lanes, then kNumCounters counters. One copy per GPU, held until the process
// ends; every call on a GPU uses that GPU's copy, so calls that overlap on one GPU race.
// The array's name is only a host-side handle: its device address is asked for once per device and kept.
// ---------------------------------------------------------------------------------------------------
constexpr uint64_t kNoAllocStackBytes = sizeof(int32_t) * (uint64_t)kOverflowStack * kNoAllocLanes;
constexpr uint64_t kNoAllocBytes = kNoAllocStackBytes + sizeof(unsigned long long) * kNumCounters;
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

constexpr int kMaxScratchDevices = 64;
static void* g_scratch_ptr[kMaxScratchDevices] = {nullptr};
static std::mutex g_scratch_mu;  // contexts may be created from several threads at once

hipError_t noAllocScratch(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxScratchDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur != device) return hipErrorInvalidDevice;  // the address belongs to the device that is current
        void* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// `scratch_ptr` = what noAllocScratch returned for the current device; rays and hits are device memory.
// Enqueues: clear of the counters, the intersection kernel on kNoAllocLanes lanes, copy of the counters to
// `host_counters` (pinned, kNumCounters words). Allocates nothing.
hipError_t launchIntersectBatchNoAlloc(const DScene& S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits, void* scratch_ptr,
                                       uint32_t ovf_cap, unsigned long long* host_counters, hipStream_t s) {
    int32_t* ovf = reinterpret_cast<int32_t*>(scratch_ptr);
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(scratch_ptr) + kNoAllocStackBytes);
    hipError_t e = hipMemsetAsync(counters, 0, sizeof(unsigned long long) * kNumCounters, s);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)std::min<uint64_t>(kNoAllocLanes / kBlock, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_intersect_batch, dim3(blocks), dim3(kBlock), 0, s, S, rays, n, hits, ovf, (uint64_t)kNoAllocLanes, ovf_cap, counters);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(host_counters, counters, sizeof(unsigned long long) * kNumCounters, hipMemcpyDeviceToHost, s);
}

}  // namespace ub

#ifdef UBDPT_WAVE_TIMES
extern "C" void ubdpt_debug_wave_times(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ub::g_wave_times), sizeof(unsigned long long) * 16 * 12);
    (void)hipMemcpyFromSymbol(out + 16 * 12, HIP_SYMBOL(ub::g_wave_phase), sizeof(unsigned long long) * 8);
    (void)hipMemcpyFromSymbol(out + 16 * 12 + 8, HIP_SYMBOL(ub::g_wave_lanes), sizeof(unsigned long long) * 4);
}
#endif
