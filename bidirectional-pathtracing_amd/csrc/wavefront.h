// wavefront.h — eye and light subpath generation as a wavefront pipeline (included by render.hip).
//
// computeEyeSubpath / computeLightSubpath (uengine/ubdptrenderer.cpp:224-594) are loops of
// "intersect, decide, scatter, push". Run one loop per lane and a wave idles until its longest path
// ends. Here every bounce is one launch over the compacted queue of live subpaths:
//
//   k_wf_gen     primary rays (camera, :231-257) and emission rays (:413-454) for the whole batch
//   k_wf_trace   closest hit of every live subpath's ray (wavefront_split.h)
//   k_wf_shade   per workgroup chunk of 256 traced subpaths: lobe choice (uapp/material.h), counting sort of
//                the chunk by (eye|light, lobe) in LDS, then Russian roulette and pdf bookkeeping for the hit,
//                s=0 emission sample, BSDF sampling, vertex push, next ray; survivors are appended to the
//                next queue with ONE global atomic per chunk (wavefront_split.h)
//   k_wf_tail    after the wavefront bounces the few survivors (long specular chains) finish in a per-lane loop
//
// Subpath id: [0, N) are the eye subpaths, [N, 2N) the light subpaths of the batch's N pixel-samples.
#pragma once

namespace ub {

// What a subpath keeps between two bounces: everything the next bounce needs of the vertices already
// stored, so that subpath generation only ever WRITES vertex records. Two 64-byte lines; the ray
// tracing stage reads the first one only (its eighth double is the subpath's count word).
struct Carry {
    d3 pos;          // last stored vertex = origin of the ray in flight
    d3 dir;          // its direction
    double p_psa;    // pdf of the scattering that chose dir (light subpath before its first hit: dirT.y of the emission sample)
    d3 a;            // throughput the next vertex receives
    d3 Ns;           // shading normal of the last stored vertex
    double q;        // roulette survival probability the ray in flight already passed
    double pending;  // area pdf the vertex BEFORE the last one receives if the ray in flight finds a
                     // surface to continue from (the reference writes it only then, :379-395 / :568-584)
};

// `primary_eye`: an eye subpath before its first bounce. The second line of its record would hold constants
// (throughput 1, the lens normal, no roulette, nothing pending); k_wf_gen does not write them and they are made here.
DEV void loadCarry(const RenderArgs& A, uint64_t id, Carry& c, bool primary_eye) {
    const double* p = A.carry + id * kCarryDoubles;
    c.pos = mk3(p[0], p[1], p[2]);
    c.dir = mk3(p[3], p[4], p[5]);
    c.p_psa = p[6];
    if (primary_eye) {
        c.a = mk3(1.0, 1.0, 1.0);
        c.Ns = transformVector(A.S.invV, mk3(0, 0, 1));
        c.q = 1.0;
        c.pending = 0;
    } else {
        c.a = mk3(p[8], p[9], p[10]);
        c.Ns = mk3(p[11], p[12], p[13]);
        c.q = p[14];
        c.pending = p[15];
    }
}
DEV void storeCarryRay(const RenderArgs& A, uint64_t id, const Carry& c) {  // first line only
    double* p = A.carry + id * kCarryDoubles;
    p[0] = c.pos.x; p[1] = c.pos.y; p[2] = c.pos.z;
    p[3] = c.dir.x; p[4] = c.dir.y; p[5] = c.dir.z;
    p[6] = c.p_psa;
}
DEV void loadCarryRay(const RenderArgs& A, uint64_t id, Ray& r) {
    const double* p = A.carry + id * kCarryDoubles;
    r.o = mk3(p[0], p[1], p[2]);
    r.d = mk3(p[3], p[4], p[5]);
}
DEV void storeCarry(const RenderArgs& A, uint64_t id, const Carry& c) {
    double* p = A.carry + id * kCarryDoubles;
    st2(p + 0, c.pos.x, c.pos.y); st2(p + 2, c.pos.z, c.dir.x); st2(p + 4, c.dir.y, c.dir.z);
    p[6] = c.p_psa;   // (p[7] holds the count word)
    st2(p + 8, c.a.x, c.a.y); st2(p + 10, c.a.z, c.Ns.x); st2(p + 12, c.Ns.y, c.Ns.z); st2(p + 14, c.q, c.pending);
}
DEV void storeHit(const RenderArgs& A, uint64_t id, const Closest& h) {
    double* p = A.wf_hit + id * kHitDoubles;
    st2(p + 0, h.d, h.u);
    st2(p + 2, h.v, __longlong_as_double((long long)(((unsigned long long)h.face << 32) | (uint32_t)h.object)));
}
DEV void loadHit(const RenderArgs& A, uint64_t id, Closest& h) {
    const double* p = A.wf_hit + id * kHitDoubles;
    h.d = p[0]; h.u = p[1]; h.v = p[2];
    unsigned long long w = (unsigned long long)__double_as_longlong(p[3]);
    h.object = (int32_t)(uint32_t)w; h.face = (uint32_t)(w >> 32);
}

// Appends the lanes with pred == true to a queue: one ballot, one atomic per wave.
DEV uint32_t waveReserve(bool pred, uint32_t* counter) {
    unsigned long long m = __ballot(pred);
    if (m == 0) return 0;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
}

constexpr int kWfRegions = 8;   // (light ? 4 : 0) + lobe + 1

DEV Rng pathRng(const RenderArgs& A, uint64_t i, bool light, uint64_t& pixel) {
    uint64_t pass;
    uint32_t px, py;
    decodePath(A, i, pass, pixel, px, py);
    Rng r;
    r.key = pathKey(A.seed, pass, pixel);
    r.stream = light ? STREAM_LIGHT : STREAM_EYE;
    return r;
}

// One bounce of one subpath after the ray described by `cy` found `hit`, with the lobe `lobe` already
// chosen: the second half of one loop iteration of computeEyeSubpath / computeLightSubpath followed
// by the first half of the next. The tests of the next iteration that do not depend on what the next
// ray hits (throughput and pdf positive, :346 / :537, and Russian roulette — the generator is
// counter-based, so the draw does not depend on when it is made) are applied here already: a subpath
// that would fail them is not traced again, and what the reference would compute from the last TWO
// vertices after the next hit is computed now, while both are in registers, and carried in `cy`.
// Returns true when a vertex was pushed and the subpath goes on; `cy` then describes the next ray.
// n = subpath.size() and meta = the subpath's count word on entry, both updated on push.
template <bool STATS, bool LIGHT>
DEV bool shadeStep(const RenderArgs& A, uint64_t i, uint64_t pixel, const Rng& rng, uint32_t& n, uint32_t& meta,
                   const Closest& hit, int lobe, Carry& cy, Tally& ty) {
    const DScene& S = A.S;
    const PathStore& P = LIGHT ? A.L : A.E;
    uint64_t npix = (uint64_t)S.img_res_x * S.img_res_y;
    Ray ray;
    ray.o = cy.pos; ray.d = cy.dir;
    Vertex nv;
    surfacePoint(S, ray, hit, nv.sp);
    nv.sp.lobe = lobe;
    nv.p_light_A = 0; nv.p_eye_A = 0; nv.specular = false;
    Tail cur;
    cur.pos = cy.pos;
    cur.Ns = cy.Ns;
    cur.specular = !P.connectable(meta, n - 1, i);
    nv.a = cy.a;
    if (n == 1) {
        if (!LIGHT) nv.p_eye_A = 1.0 / S.image_plane_area;  // :265-267
        else {                                              // :463-472, hit still in object-local space
            double G = geomAbs(cur.pos, cur.Ns, nv.sp.pos, nv.sp.Ns);
            nv.p_light_A = (1 / (2.0 * kPi * cy.p_psa)) * G;
        }
    } else {
        P.st(LIGHT ? F_PE : F_PL, n - 2, i, cy.pending);
        if (cur.specular) {
            if (!LIGHT) nv.p_eye_A = cy.q * cy.p_psa;
            else        nv.p_light_A = cy.q * 1.0;
        } else {
            // G_next uses the new hit's OBJECT-LOCAL position and normal, as the reference does (:389-393, :578-582)
            double G_next = geomAbs(cur.pos, cur.Ns, nv.sp.pos, nv.sp.Ns);
            if (!LIGHT) nv.p_eye_A = cy.q * cy.p_psa * G_next;
            else        nv.p_light_A = cy.q * cy.p_psa * G_next;
        }
    }
    const DObject& obj = S.objects[nv.sp.object];
    const bool emits = !LIGHT && obj.is_emitter;  // s = 0 sample: the eye subpath ran into a light
    bool prev_specular = false;
    if (emits) {
        cur.p_eye_A = P.ld(F_PE, n - 1, i);
        prev_specular = n > 1 && !P.connectable(meta, n - 2, i);
    }
    if (lobe == LOBE_NONE) {
        if (emits) {  // :273-283
            nv.sp.pos = transformPoint(obj.W, nv.sp.pos);
            nv.sp.Ns = transformVectorT(obj.invW, nv.sp.Ns);
            d3 I = s0sample(S, P, i, n, cur, prev_specular, nv);
            if (I.x != 0 || I.y != 0 || I.z != 0) { fbAdd(A.fb, npix, pixel + fbSlotOffset(A, i), I); if (STATS) ty.fb++; }
        }
        return false;
    }
    if (n >= S.max_path_vertices) { if (STATS) ty.capped++; return false; }
    d3 lTs = nv.sp.Ts, lNs = nv.sp.Ns, lBs = nv.sp.Bs;
    Frame fr; fr.Ng = nv.sp.Ng; fr.Ns = nv.sp.Ns; fr.Ts = nv.sp.Ts; fr.Bs = nv.sp.Bs;
    d3 w = transformVector(obj.invW, -ray.d);
    d3 tex = textureSample(S, obj.texture, nv.sp.tex_u, nv.sp.tex_v);
    nv.tex = tex;
    double p_psa;
    d3 fs, dirT;
    if (!bsdfScatter(lobe, obj.mat_a, tex, fr, w, rng(hitDim(n, DIM_SCATTER_1)), rng(hitDim(n, DIM_SCATTER_2)), dirT,
                     p_psa, fs, nv.specular))
        return false;
    finishVertex(obj, nv, dirT);
    if (emits) {
        d3 I = s0sample(S, P, i, n, cur, prev_specular, nv);
        if (I.x != 0 || I.y != 0 || I.z != 0) { fbAdd(A.fb, npix, pixel + fbSlotOffset(A, i), I); if (STATS) ty.fb++; }
    }
    if (!P.openSlot(n, i)) { poolExhausted(A); return false; }
    storeVertex(P, n, i, nv);
    n++;
    meta = ((meta & 0xffffff00u) | specularBit(n - 1, nv.specular)) | n;
    P.setMeta(i, meta);
    // the head of the next loop iteration (:346-356 / :537-547), n being the new size
    double fs_sum = fs.x + fs.y + fs.z;
    if (!(p_psa > 0) || !(fs_sum > 0)) return false;
    double q;
    if (n < S.min_depth) q = 1.0;
    else {
        q = fmin(1.0, (fs_sum / 3.0) / p_psa);
        if (rng(hitDim(n - 1, DIM_ROULETTE)) > q) return false;
    }
    // throughput of the next vertex and the pdf its hit will hand back to vertex n - 2 (:358-395 / :549-584)
    if (nv.specular) {
        cy.a = nv.a * fs;
        cy.pending = !LIGHT ? q * p_psa : q * 1.0;
    } else {
        d3 a = fs / p_psa;
        cy.a = nv.a * a;
        double G_prev = !LIGHT ? geomAbs(nv.sp.pos, nv.sp.Ns, cur.pos, cur.Ns) : geomAbs(cur.pos, cur.Ns, nv.sp.pos, nv.sp.Ns);
        cy.pending = q * p_psa * G_prev;
    }
    cy.pos = nv.sp.pos;
    cy.Ns = nv.sp.Ns;
    cy.dir = transformVector(obj.W, float_tnb_mul(lTs, lNs, lBs, dirT));
    cy.p_psa = p_psa;
    cy.q = q;
    return true;
}

// Records of consecutive subpaths lie side by side in memory. A lane that writes its own record with 16-byte stores
// makes every store instruction touch 64 different lines; measured on this part a kernel that only stores 192-byte
// records gets 3.7 TB/s that way, 5.6 TB/s when the wave stages the records in LDS and writes them out so that
// consecutive lanes store consecutive 16-byte pieces, 6.9 TB/s for a plain streaming store (tools/store_bench.hip).
// waveStore: every lane of the wave hands in PIECES pieces of its record; the record of lane l starts at
// dst0 + l * stride_doubles + 2 * first_piece doubles. All 64 lanes must take part.
constexpr int kStagePieces = 4;
template <int PIECES>
DEV void waveStore(v2d* stage, int lane, const v2d (&piece)[PIECES], double* dst0, uint32_t stride_doubles, int first_piece) {
    static_assert(PIECES <= kStagePieces, "stage too small");
#pragma unroll
    for (int k = 0; k < PIECES; k++) stage[lane * PIECES + k] = piece[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const int p = k * 64 + lane;
        const int r = p / PIECES, q = p - r * PIECES;
        *reinterpret_cast<v2d*>(dst0 + (uint64_t)r * stride_doubles + 2 * (first_piece + q)) = stage[p];
    }
    __builtin_amdgcn_wave_barrier();
}
DEV v2d mk2(double a, double b) { v2d v; v.x = a; v.y = b; return v; }
// The lanes with `on` set hand in PIECES pieces each, which go to consecutive records of PIECES pieces from dst0 on, in
// lane order (the records of a queue whose places were handed out by a ballot). Every lane of the wave must call.
template <int PIECES>
DEV void waveStoreRanked(v2d* stage, int lane, bool on, const v2d (&piece)[PIECES], double* dst0) {
    static_assert(PIECES <= kStagePieces, "stage too small");
    const unsigned long long m = __ballot(on);
    const int rank = __popcll(m & ((1ULL << lane) - 1ULL)), count = __popcll(m);
    if (on) {
#pragma unroll
        for (int k = 0; k < PIECES; k++) stage[rank * PIECES + k] = piece[k];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const int p = k * 64 + lane;
        if (p < count * PIECES) *reinterpret_cast<v2d*>(dst0 + 2 * p) = stage[p];
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- generation ---------------------------------------------------------------------------------------
// Vertex 0 and the first ray of every subpath (eye: lens vertex and primary ray, :231-257; light: emitter vertex and
// emission ray, :413-454): stores the vertex, the count word and the carry record.
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_wf_gen(RenderArgs A) {
    __shared__ v2d s_stage[kBlock / 64][64 * kStagePieces];
    uint64_t id = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const DScene& S = A.S;
    uint64_t N = A.E.n;
    const int lane = threadIdx.x & 63;
    v2d* stage = s_stage[threadIdx.x >> 6];
    bool live = false;
    const bool light = id >= N;
    const uint64_t i = light ? id - N : id;
    Carry cy;
    cy.pos = cy.dir = cy.a = cy.Ns = mk3(0, 0, 0);
    cy.p_psa = 0; cy.q = 1.0; cy.pending = 0;
    Vertex v0;   // lens or emitter vertex
    if (id < 2 * N) {
        uint64_t pass, pixel;
        uint32_t px, py;
        decodePath(A, i, pass, pixel, px, py);
        Rng rng{pathKey(A.seed, pass, pixel), light ? (uint32_t)STREAM_LIGHT : (uint32_t)STREAM_EYE};
        if (!light) {
            uint32_t pixel_sample = (uint32_t)(pass % S.num_pixel_strata);
            uint32_t lens_sample = (uint32_t)(pass % S.num_lens_strata);
            d2 disk = sampleUnitDiskStratified(S.num_lens_strata, lens_sample, rng(DIM_LENS_X), rng(DIM_LENS_Y));
            d3 lens_pointV = mk3(S.lens_radius * disk.x, S.lens_radius * disk.y, 0);
            Vertex& lens = v0;
            lens.a = mk3(1.0, 1.0, 1.0);
            lens.tex = mk3(0, 0, 0);
            lens.sp.pos = mul_xyz(S.invV, lens_pointV, 1.0);
            lens.sp.Ns = lens.sp.Ng = transformVector(S.invV, mk3(0, 0, 1));
            lens.sp.Ts = transformVector(S.invV, mk3(1, 0, 0));
            lens.sp.Bs = transformVector(S.invV, mk3(0, 1, 0));
            lens.sp.tex_u = 0; lens.sp.tex_v = 0; lens.sp.object = -1; lens.sp.lobe = LOBE_NONE; lens.sp.face = 0;
            lens.p_eye_A = 1.0 / S.lens_area;  // lens_area is 0 in the reference -> +inf
            lens.p_light_A = 0;
            lens.specular = false;
            d2 pp = sampleUnitRectStratified(S.num_pixel_strata, pixel_sample, rng(DIM_PIXEL_X), rng(DIM_PIXEL_Y));
            d3 image_pointV = mk3(-S.image_plane_ratio + ((double)px + pp.x) * S.pixel_width,
                                  1.0 - ((double)py + pp.y) * S.pixel_height, S.image_plane_distance);
            image_pointV = normalize(image_pointV);
            d3 focus_plane_pointV = image_pointV * (S.focus_plane_distance / image_pointV.z);
            cy.dir = transformVector(S.invV, focus_plane_pointV - lens_pointV);
            cy.pos = lens.sp.pos; cy.Ns = lens.sp.Ns; cy.a = lens.a;
            live = true;
        } else if (S.num_emitters > 0) {
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
            Vertex& ev = v0;
            emitterRandomPoint(S, em, rng, ev.sp.pos, ev.sp.Ng, ev.sp.Ns, ev.sp.Ts, ev.sp.Bs);
            ev.sp.tex_u = 0; ev.sp.tex_v = 0; ev.sp.object = ei; ev.sp.lobe = LOBE_NONE; ev.sp.face = 0;
            ev.a = em.power;
            ev.tex = mk3(0, 0, 0);
            ev.p_light_A = em.prob * (1.0 / em.area);
            ev.p_eye_A = 0;
            ev.specular = false;
            d3 dirT = samplePosHemUniform(rng(DIM_EMIT_ANGLE), rng(DIM_EMIT_V));
            cy.dir = float_tnb_mul(ev.sp.Ts, ev.sp.Ns, ev.sp.Bs, dirT);
            cy.p_psa = dirT.y;
            cy.pos = ev.sp.pos; cy.Ns = ev.sp.Ns; cy.a = ev.a;
            live = true;
        }
    }
    // Stores: vertex 0, the count word and the carry record. A wave whose 64 lanes are all live subpaths of one kind (all
    // but the wave at the border between the eye and the light subpaths and the last one) writes them together (waveStore),
    // when the count words live inside the carry records (their eighth double), as they do in this pipeline.
    const PathStore& P = light ? A.L : A.E;
    const bool count_in_carry = reinterpret_cast<const double*>(P.count) == A.carry + (light ? N * kCarryDoubles : 0) + kCarryCountSlot &&
                                P.count_stride == kCarryDoubles * 2;
    const bool together = __all(live && count_in_carry) && __all(light) == __any(light);
    if (together) {
        const uint64_t i0 = i - (uint64_t)lane;
        double of;
        {
            int32_t w[2] = {v0.sp.object, ((v0.sp.lobe + 1) & 0xff)};
            __builtin_memcpy(&of, w, 8);
        }
        const double meta1 = __longlong_as_double(1LL);   // count word 1 (one vertex, not specular) in the low half
        double* vrec = P.rp(0, i0);
        double* crec = A.carry + (id - (uint64_t)lane) * kCarryDoubles;
        {   // line 0 of the vertex record, as storeVertex / storeLensVertex lay it out
            v2d pc[4] = {mk2(v0.sp.pos.x, v0.sp.pos.y), mk2(v0.sp.pos.z, v0.p_light_A), mk2(v0.p_eye_A, of), mk2(0.0, 0.0)};
            waveStore<4>(stage, lane, pc, vrec, kVertexDoubles, 0);
        }
        if (light) {
            {
                v2d pc[4] = {mk2(v0.tex.z, v0.sp.Ng.x), mk2(v0.sp.Ng.y, v0.sp.Ng.z), mk2(v0.sp.Ns.x, v0.sp.Ns.y), mk2(v0.sp.Ns.z, v0.sp.Ts.x)};
                waveStore<4>(stage, lane, pc, vrec, kVertexDoubles, 4);
            }
            {
                v2d pc[4] = {mk2(v0.sp.Ts.y, v0.sp.Ts.z), mk2(v0.sp.Bs.x, v0.sp.Bs.y), mk2(v0.sp.Bs.z, v0.a.x), mk2(v0.a.y, v0.a.z)};
                waveStore<4>(stage, lane, pc, vrec, kVertexDoubles, 8);
            }
        }
        {   // line 0 of the carry record, count word included
            v2d pc[4] = {mk2(cy.pos.x, cy.pos.y), mk2(cy.pos.z, cy.dir.x), mk2(cy.dir.y, cy.dir.z), mk2(cy.p_psa, meta1)};
            waveStore<4>(stage, lane, pc, crec, kCarryDoubles, 0);
        }
        if (light) {   // (the rest of an eye subpath's first record is constant, see loadCarry)
            v2d pc[4] = {mk2(cy.a.x, cy.a.y), mk2(cy.a.z, cy.Ns.x), mk2(cy.Ns.y, cy.Ns.z), mk2(cy.q, cy.pending)};
            waveStore<4>(stage, lane, pc, crec, kCarryDoubles, 4);
        }
        return;
    }
    if (id >= 2 * N) return;
    if (!light) {
        storeLensVertex(A.E, i, v0);
        A.E.setMeta(i, 1);
        storeCarryRay(A, id, cy);
    } else if (live) {
        storeVertex(A.L, 0, i, v0);
        A.L.setMeta(i, 1);
        storeCarry(A, id, cy);
    } else {
        A.L.setMeta(i, 0);
    }
}

// ---- tail: survivors of the last wavefront bounce finish in a per-lane loop ---------------------------
#ifndef UBDPT_TAIL_WAVES
#define UBDPT_TAIL_WAVES 2
#endif
template <bool STATS, int LDS_LEVEL>
__global__ __launch_bounds__(kBlock, UBDPT_TAIL_WAVES) void k_wf_tail(RenderArgs A0, uint32_t depth) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ DObject s_objects[ldsObjects(LDS_LEVEL)];
    __shared__ LdsNode s_nodes[ldsNodes(LDS_LEVEL)];
    __shared__ Prim s_prims[ldsPrims(LDS_LEVEL)];
    RenderArgs A = A0;
    if (LDS_LEVEL >= 1) A.S.objects = stageObjects(A0.S, s_objects);
    if (LDS_LEVEL >= 2) stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    uint64_t N = A.E.n;
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    uint32_t total = A.wf_count[(uint64_t)depth * kWfQStride];
    const uint32_t* q = A.wf_trace_q + (uint64_t)(depth & 1) * 2 * N;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    for (uint64_t j = gid; j < total; j += nthreads) {
        uint32_t id = q[j];
        bool light = id >= N;
        uint64_t i = light ? id - N : id;
        const PathStore& P = light ? A.L : A.E;
        uint32_t meta = P.meta(i);
        uint32_t n = meta & 0xffu;
        uint64_t pixel;
        Rng rng = pathRng(A, i, light, pixel);
        Carry cy;
        loadCarry(A, id, cy, !light && n == 1);
        for (;;) {
            Ray ray;
            ray.o = cy.pos; ray.d = cy.dir;
            Closest hit = closestHit<STATS, (LDS_LEVEL >= 2)>(S, ray, st, ty);
            if (hit.object < 0) break;
            int lobe = pickLobe(S.objects[hit.object], rng(hitDim(n, DIM_LOBE)));
            if (light && lobe == LOBE_NONE) break;
            bool cont = light ? shadeStep<STATS, true>(A, i, pixel, rng, n, meta, hit, lobe, cy, ty)
                              : shadeStep<STATS, false>(A, i, pixel, rng, n, meta, hit, lobe, cy, ty);
            if (!cont) break;
        }
        (light ? A.fin_L : A.fin_E)[i] = meta;  // the subpath is complete
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

}  // namespace ub
