// wavefront_split.h — the wavefront stages as SMALL kernels (included by render.hip after wavefront.h).
//
// These kernels are bound by memory latency times occupancy: a bounce is a chain of ~25 dependent
// memory phases, and a kernel that holds the traversal state AND the fp64 shading state needs
// >200 VGPRs (2 waves per SIMD). Splitting at the natural seams keeps each kernel's live state small:
//
//   k_wf_trace     ray in, (object, face, d, u, v, lobe) out; the 256 subpaths of a chunk are ordered by
//                  (eye|light, lobe) with a stable counting sort in LDS and appended to the hit
//                  queue as one block: one global atomic per chunk, chunks padded to whole waves so
//                  that a wave of the shade kernel sees one lobe (material-sorted queue)
//   k_wf_shade     no traversal: pdf bookkeeping, s=0 sample, BSDF sampling, vertex push, next ray;
//                  survivors -> next trace queue (one atomic per chunk)
//   k_pair_visible the rejecting half of a connection, ending with the shadow ray; survivors -> queue
//   k_pair_eval    connection factor, MIS weight, accumulation for the visible pairs only
#pragma once

namespace ub {

constexpr uint32_t kQueueSentinel = 0xffffffffu;

// Waves per SIMD the ray-tracing kernels are compiled for (register budget = 512 / this; 1 = whatever the
// compiler needs, which comes to 4). A scene that lives in LDS leaves room for 4 workgroups per CU and its rays
// are bound by arithmetic. Rays that walk a tree in HBM wait on dependent node fetches: for them a fifth wave per
// SIMD is worth the handful of spilled registers it costs (+7 % on scenes of 100 k and 2 M triangles).
#ifndef UBDPT_TRACE_WAVES
#define UBDPT_TRACE_WAVES 1
#endif
#ifndef UBDPT_TRACE_WAVES_HBM
#define UBDPT_TRACE_WAVES_HBM 5
#endif
constexpr int traceWaves(int lds_level) { return lds_level >= 2 ? UBDPT_TRACE_WAVES : UBDPT_TRACE_WAVES_HBM; }
// the same for the shading and the connection evaluation kernel
#ifndef UBDPT_SHADE_WAVES
#define UBDPT_SHADE_WAVES 1
#endif
#ifndef UBDPT_EVAL_WAVES
#define UBDPT_EVAL_WAVES 1
#endif

// (Workgroup b runs on XCD b % 8, each with its own L2. Remapping the chunks so that one XCD sweeps a contiguous
// eighth of a queue — a band of the image for primary rays — was measured on scenes of 14, 100 k and 2 M
// primitives: no difference, the rays that dominate are incoherent. The plain mapping stays.)

// Appends the lanes with pred to a queue with ONE global atomic per workgroup. Every lane of the
// workgroup must call it. Returns the lane's slot (valid when pred).
DEV uint32_t blockAppend(bool pred, uint32_t* counter, uint32_t* s_wave, uint32_t* s_base) {
    const uint32_t tid = threadIdx.x;
    unsigned long long m = __ballot(pred);
    if ((tid & 63) == 0) s_wave[tid >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) tot += s_wave[w];
        *s_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = *s_base + (uint32_t)__popcll(m & ((1ULL << (tid & 63)) - 1ULL));
    for (uint32_t w = 0; w < (tid >> 6); w++) pos += s_wave[w];
    __syncthreads();  // s_wave / s_base may be reused by the next call
    return pos;
}

// ---- trace ------------------------------------------------------------------------------------------------
// round 0 reads the trace queue of this depth and gives every ray `budget` traversal steps; a ray that runs out of
// them leaves its closest candidate so far in its hit record and its id in the deferred queue, which round 1
// (budget 0xffffffff) finishes: the long rays together instead of one per wave holding 63 idle lanes.
// budget 0xffffffff in round 0: no ray is deferred and there is no round 1.
// (Generating the subpaths inside the first launch of this kernel instead of in k_wf_gen was measured: the generation
// code costs the kernel a quarter of its occupancy, inlined or not, and the whole is slower: 411 against 418 Mpaths/s.)
template <bool STATS, int LDS_LEVEL>
__global__ __launch_bounds__(kBlock, traceWaves(LDS_LEVEL)) void k_wf_trace(RenderArgs A0, uint32_t depth, uint32_t round,
                                                                         uint32_t budget) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ uint32_t s_wcnt[kWfRegions][kBlock / 64];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ DObject s_objects[ldsObjects(LDS_LEVEL)];
    __shared__ LdsNode s_nodes[ldsPrims(LDS_LEVEL)];
    __shared__ Prim s_prims[ldsPrims(LDS_LEVEL)];
    RenderArgs A = A0;
    if (LDS_LEVEL >= 1) A.S.objects = stageObjects(A0.S, s_objects);
    if (LDS_LEVEL >= 2) stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    const uint64_t N = A.E.n;
    const uint64_t n2 = 2 * N;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + tid;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const bool first = round == 0;
    uint32_t* defer_cnt = A.wf_count + (uint64_t)depth * kWfQStride + 2;
    const uint64_t total = !first ? (uint64_t)*defer_cnt : depth == 0 ? n2 : (uint64_t)A.wf_count[(uint64_t)depth * kWfQStride];
    const uint32_t* q = first ? A.wf_trace_q + (uint64_t)(depth & 1) * n2 : A.wf_defer_q;
    uint32_t* hit_cnt = A.wf_count + (uint64_t)depth * kWfQStride + 1;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < total; base += nthreads) {
        uint64_t j = base + tid;
        int region = kWfRegions;  // dead
        uint32_t id = 0;
        bool deferred = false;
        if (j < total) {
            id = (first && depth == 0) ? (uint32_t)j : q[j];
            bool light = id >= N;
            uint64_t i = light ? id - N : id;
            const PathStore& P = light ? A.L : A.E;
            uint32_t n = P.cnt(i);
            if (n > 0) {
                Ray ray;
                loadCarryRay(A, id, ray);
                Closest hit;
                if (first) {
                    hit.d = __builtin_inf(); hit.u = 0; hit.v = 0; hit.object = -1; hit.face = 0;
                    if (STATS) ty.closest++;
                } else {
                    loadHit(A, id, hit);  // the closest candidate round 0 had found when its budget ran out
                }
                if (S.accel == 1) hit = closestHitLinear(S, ray, STATS ? &ty.tc : nullptr);
                else deferred = !closestHitBvhBudget(S, ray, st, STATS ? &ty.tc : nullptr, budget, hit);
                if (deferred) {
                    storeHit(A, id, hit);
                } else if (hit.object >= 0) {
                    uint64_t pixel;
                    Rng rng = pathRng(A, i, light, pixel);
                    int lobe = pickLobe(S.objects[hit.object], rng(hitDim(n, DIM_LOBE)));
                    if (!(light && lobe == LOBE_NONE)) {  // a light subpath ends at an absorbing hit (:460, :527)
                        region = (light ? 4 : 0) + lobe + 1;
                        storeHit(A, id, hit);
                    }
                }
            }
        }
        if (budget != 0xffffffffu) {  // uniform
            uint32_t dpos = blockAppend(deferred, defer_cnt, s_wave, &s_base);
            if (deferred) A.wf_defer_q[dpos] = id;
        }
        // stable counting sort of the chunk by region: rank inside the wave by ballot, wave totals in LDS
        uint32_t rank = 0;
#pragma unroll
        for (int r = 0; r < kWfRegions; r++) {
            unsigned long long m = __ballot(region == r);
            if (lane == 0) s_wcnt[r][wave] = (uint32_t)__popcll(m);
            if (region == r) rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        }
        __syncthreads();
        uint32_t slot = rank, live = 0;
#pragma unroll
        for (int r = 0; r < kWfRegions; r++)
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) {
                uint32_t c = s_wcnt[r][w];
                live += c;
                if (r < region || (r == region && (uint32_t)w < wave)) slot += c;
            }
        const uint32_t padded = (live + 63u) & ~63u;  // whole waves: a shade wave never straddles two chunks
        if (tid == 0) s_base = padded ? atomicAdd(hit_cnt, padded) : 0u;
        __syncthreads();
        if (region < kWfRegions) A.wf_hit_q[s_base + slot] = id | ((uint32_t)region << 29);
        if (tid < padded - live) A.wf_hit_q[s_base + live + tid] = kQueueSentinel;
        __syncthreads();
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// ---- shade ------------------------------------------------------------------------------------------------
// (Keeping the corner positions and shading attributes of the faces of a small scene in LDS as well removes the
// last dependent global fetch between a hit and its stores; measured: no difference, the kernel is bound by its
// stores.)
template <bool STATS, uint32_t LDS_OBJECTS>
__global__ __launch_bounds__(kBlock, UBDPT_SHADE_WAVES) void k_wf_shade(RenderArgs A0, uint32_t depth) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ DObject s_objects[LDS_OBJECTS ? LDS_OBJECTS : 1];
    RenderArgs A = A0;
    if (LDS_OBJECTS) A.S.objects = stageObjects(A0.S, s_objects);
    const uint64_t N = A.E.n;
    const uint64_t n2 = 2 * N;
    const uint32_t tid = threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint64_t total = (uint64_t)A.wf_count[(uint64_t)depth * kWfQStride + 1];
    uint32_t* next_cnt = A.wf_count + (uint64_t)(depth + 1) * kWfQStride;
    uint32_t* next_q = A.wf_trace_q + (uint64_t)((depth + 1) & 1) * n2;
    Tally ty;
    for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < total; base += nthreads) {
        uint64_t j = base + tid;
        bool cont = false;
        uint32_t id = 0;
        if (j < total) {
            uint32_t e = A.wf_hit_q[j];
            if (e != kQueueSentinel) {
                id = e & 0x1fffffffu;
                uint32_t r = e >> 29;
                bool light = r >= 4;
                int lobe = (int)(r & 3) - 1;
                uint64_t i = light ? id - N : id;
                const PathStore& P = light ? A.L : A.E;
                uint32_t n = P.cnt(i);
                uint64_t pixel;
                Rng rng = pathRng(A, i, light, pixel);
                Carry cy;
                loadCarry(A, id, cy, !light && n == 1);
                Closest hit;
                loadHit(A, id, hit);
                cont = light ? shadeStep<STATS, true>(A, i, pixel, rng, n, hit, lobe, cy, ty)
                             : shadeStep<STATS, false>(A, i, pixel, rng, n, hit, lobe, cy, ty);
                if (cont) storeCarry(A, id, cy);
            }
        }
        uint32_t pos = blockAppend(cont, &next_cnt[0], s_wave, &s_base);
        if (cont) next_q[pos] = id;
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
}

// ---- connections: visibility, then evaluation -----------------------------------------------------------
// round 0 reads the pair queue, round r > 0 the deferred queue (r - 1) & 1. Every shadow ray gets a budget of
// `budget` traversal steps; the pairs whose ray runs out of it go to the deferred queue r & 1 for the next round
// (budget 0xffffffff: no ray is ever deferred; the last round must be launched with it).
template <bool STATS, int LDS_LEVEL>
__global__ __launch_bounds__(kBlock, traceWaves(LDS_LEVEL)) void k_pair_visible(RenderArgs A0, uint32_t round, uint32_t budget) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ DObject s_objects[ldsObjects(LDS_LEVEL)];
    __shared__ LdsNode s_nodes[ldsPrims(LDS_LEVEL)];
    __shared__ Prim s_prims[ldsPrims(LDS_LEVEL)];
    RenderArgs A = A0;
    if (LDS_LEVEL >= 1) A.S.objects = stageObjects(A0.S, s_objects);
    if (LDS_LEVEL >= 2) stageTree(A.S, s_nodes, s_prims);
    const uint32_t tid = threadIdx.x;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + tid;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const bool first = round == 0;
    // (no dynamic index into the local copy A: it would be moved to scratch memory as a whole)
    const bool odd = round & 1;
    uint64_t* defer_q = odd ? A0.defer_pairs[1] : A0.defer_pairs[0];
    uint32_t* defer_n = odd ? A0.defer_count[1] : A0.defer_count[0];
    const uint64_t* prev_q = odd ? A0.defer_pairs[0] : A0.defer_pairs[1];
    const uint32_t* prev_n = odd ? A0.defer_count[0] : A0.defer_count[1];
    unsigned long long total = first ? *A.pair_count : (unsigned long long)*prev_n;
    if (total > A.pair_capacity) total = A.pair_capacity;
    const uint64_t* queue = first ? A.pairs : prev_q;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < total; base += nthreads) {
        unsigned long long j = base + tid;
        int r = kBlocked;
        uint64_t p = 0;
        if (j < total) {
            p = queue[j];
            uint32_t s = (uint32_t)((p >> 8) & 0xff), t = (uint32_t)(p & 0xff);
            if (s != 0) r = pairVisibleBudget<STATS>(A, p >> 16, s, t, st, ty, budget, first);
        }
        uint32_t pos = blockAppend(r == kVisible, A.vis_count, s_wave, &s_base);
        if (r == kVisible) A.vis_pairs[pos] = p;
        if (budget != 0xffffffffu) {  // uniform; a deferred queue holds as many entries as the pair queue
            pos = blockAppend(r == kOutOfBudget, defer_n, s_wave, &s_base);
            if (r == kOutOfBudget) defer_q[pos] = p;
        }
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// Adjacent lanes mostly hold pairs of one pixel-sample that land in the same pixel (the queue is in
// pixel-sample order, k_pair_enum emits the own-pixel pairs of a pixel-sample next to each other):
// runs of equal target pixels are summed with shuffles and the first lane of a run issues the three
// atomics. Atomic throughput, not arithmetic, limits this kernel.
template <bool STATS>
__global__ __launch_bounds__(kBlock, UBDPT_EVAL_WAVES) void k_pair_eval(RenderArgs A) {
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t total = *A.vis_count;
    const uint64_t npix = (uint64_t)A.S.img_res_x * A.S.img_res_y;
    Tally ty;
    for (uint64_t base = gid - lane; base < total; base += nthreads) {  // wave-uniform trip count
        const uint64_t j = base + lane;
        uint64_t key = ~0ULL;  // target pixel, ~0: nothing to add
        d3 v = mk3(0, 0, 0);
        if (j < total) {
            uint64_t p = A.vis_pairs[j];
            uint64_t pixel;
            if (pairEval<STATS>(A, p >> 16, (uint32_t)((p >> 8) & 0xff), (uint32_t)(p & 0xff), ty, pixel, v)) key = pixel;
        }
        // runs of adjacent lanes with the same target: a run starts where the key changes
        const uint64_t prev = __shfl_up(key, 1, 64);
        const bool boundary = lane == 0 || prev != key;
        const unsigned long long starts = __ballot(boundary);
        const int run = 63 - __clzll((long long)(starts & ((2ULL << lane) - 1ULL)));  // first lane of this lane's run
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int r2 = __shfl_down(run, off, 64);
            const double x = __shfl_down(v.x, off, 64), y = __shfl_down(v.y, off, 64), z = __shfl_down(v.z, off, 64);
            if (lane + off < 64 && r2 == run) { v.x += x; v.y += y; v.z += z; }
        }
        if (boundary && key != ~0ULL) fbAdd(A.fb, npix, key, v);
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
}

}  // namespace ub
