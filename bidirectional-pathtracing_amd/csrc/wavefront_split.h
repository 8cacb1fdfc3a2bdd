// wavefront_split.h — the wavefront stages as SMALL kernels (included by render.hip after wavefront.h).
//
// These kernels are bound by memory latency times occupancy: a bounce is a chain of ~25 dependent
// memory phases, and a kernel that holds the traversal state AND the fp64 shading state needs
// >200 VGPRs (2 waves per SIMD). Splitting at the natural seams keeps each kernel's live state small:
//
//   k_wf_trace     ray in, (object, face, d, u, v) out. Persistent waves: a wave claims chunks of the trace queue and
//                  every lane that has finished its ray takes the next one while the other lanes go on walking
//   k_wf_shade     per workgroup chunk of 256 traced subpaths: lobe choice, stable counting sort of the chunk by
//                  (eye|light, lobe) in LDS so that a wave sees one BSDF (material-sorted queue), then pdf
//                  bookkeeping, s=0 sample, BSDF sampling, vertex push, next ray; survivors -> next trace queue
//   k_pair_setup   the rejecting tests of a connection that need no ray (t=1 projection, geometric-normal
//                  cosines); the shadow rays of the pairs that pass, as 64-byte records in a dense queue
//   k_rays_visible the shadow rays, persistent waves like k_wf_trace; one flag per ray
//   k_pair_compact pairs with a visible ray -> queue
//   k_pair_eval    connection factor, MIS weight, accumulation for the visible pairs only
#pragma once

namespace ub {

// Waves per SIMD the ray-tracing kernels are compiled for (a MINIMUM: it caps the registers at 512 / this; 1 = whatever
// the compiler needs). Measured on the 2 M-triangle scene with the persistent-wave kernels below: 4 waves (128
// registers, nothing spilled) 218 Mpaths/s, 5 waves (96 registers, 12..34 spilled inside the loop) 178. Fewer resident
// waves are slower (grid of 3 / 2 workgroups per CU instead of 4: 231 / 225 against 239), so occupancy does pay; it is
// the fp64 state of a ray that caps it at 4.
#ifndef UBDPT_TRACE_WAVES
#define UBDPT_TRACE_WAVES 1
#endif
#ifndef UBDPT_TRACE_WAVES_HBM
#define UBDPT_TRACE_WAVES_HBM 4
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

// ---- persistent-wave ray tracing -----------------------------------------------------------------------------
// The steps a ray needs to cross the tree are heavy-tailed: on a 2 M-triangle scene the mean shadow ray visits 9
// nodes, but the slowest of 64 rays visits 80, so a wave that traces 64 rays side by side keeps 7 of its 64
// lanes busy. Here a wave is a pool of 64 lanes fed from a queue: the wave claims chunks of kRayChunk queue
// entries (one atomic per chunk), and whenever kRefillMin lanes have finished their rays they take the next
// entries while the other lanes go on walking. A new ray's record is REQUESTED in one iteration and used in the
// next, so its load completes under the wait for the tree nodes the walking lanes fetch anyway: refilling never
// stalls the wave. Leaves are tested when kLeafBatch lanes hold one (or nothing else can make progress, or a
// lane has waited kLeafWait node steps), so the fp64 primitive test runs for many lanes at once.
// Where a wave's time goes (2 M-triangle scene, wall-clock stamps inside the loop, -DUBDPT_WAVE_TIMES build): an
// iteration takes about 4 us with 16 waves per CU; 34 % of it is spent waiting for the node and the primitive,
// 54 % in the arithmetic between the waits (node step 16 %, local ray 2 %, primitive test 12 %, bounding-sphere gate
// + pop + store 22 %), 12 % on arrivals and feeding. After the queue runs dry a wave lives another 0.2 ms on
// average (its longest ray). Measured and not kept: requesting the next node or primitive at the end of an
// iteration and using it in the next (one wait per iteration instead of two, but one step per lane and iteration
// and 12 more live registers): 212 against 239 Mpaths/s; five waves per SIMD: 12 to 34 registers spilled; testing
// leaves at once when the queue has run dry instead of waiting for a batch: no difference (three alternating runs);
// a ring of 64 rays per wave in LDS, filled ahead by LDS-DMA (global_load_lds from the carry records), out of which a
// lane that finishes takes its next ray in the same iteration: lanes per node step 37 -> 43 of 64, iterations -14 %,
// and every iteration that much longer: 247 against 251 Mpaths/s. The loop is not paced by its iteration count:
// eight-wide nodes (-40 % node steps), the ring (-14 % iterations) and cheaper slab tests (-150 instructions per node
// step, +3.5 %) all leave the time of a launch nearly where it was; half of it is the fp64 leaf phase (primitive fetch,
// triangle test, bounding-sphere gate), which none of them touches. Copying a leaf's primitive record into a per-lane
// LDS slot by LDS-DMA when the lane arrives at the leaf and testing it one iteration later (no wait for the primitive):
// this kernel 1.39 -> 1.30 ms per launch, but with 40 KB of LDS per workgroup the kernels of the other batch lane
// slowed down by more: 248 against 252 Mpaths/s. Answering the bounding-sphere gate from the hit point (well inside the
// sphere: true, else the full computation) saves two thirds of its arithmetic and gains 1 %: not worth a second,
// argued-equivalent form of a reference test. Parking up to two leaves per lane and walking on, so that the primitive
// tests (10 to 14 of 64 lanes take part in one today) run for more lanes at a time: with rounds started at 12 / 24 / 40
// parked lanes and stuck lanes waiting 3 / 6 / 10 iterations this kernel took 1.44 to 1.54 ms against 1.40.
#ifndef UBDPT_RAY_CHUNK
#define UBDPT_RAY_CHUNK 256
#endif
#ifndef UBDPT_REFILL_MIN
#define UBDPT_REFILL_MIN 16
#endif
#ifndef UBDPT_LEAF_BATCH
#define UBDPT_LEAF_BATCH 20
#endif
#ifndef UBDPT_LEAF_WAIT
#define UBDPT_LEAF_WAIT 3
#endif
constexpr uint32_t kRayChunk = UBDPT_RAY_CHUNK;
constexpr int kRefillMin = UBDPT_REFILL_MIN;
constexpr int kLeafBatch = UBDPT_LEAF_BATCH;
constexpr uint32_t kLeafWait = UBDPT_LEAF_WAIT;
enum { ST_IDLE = 0, ST_ID = 1, ST_RAY = 2, ST_ACTIVE = 3 };

// Wave-uniform view of the queue a persistent wave feeds from.
struct WaveFeed {
    uint32_t next, end;  // unread part of the chunk this wave holds
    bool drained;        // the queue has nothing left for this wave
};
// Entries a wave claims at a time: kRayChunk when the queue is long, fewer (at least 64) when that would leave
// waves of the launch without work.
DEV uint32_t chunkFor(uint32_t total) {
    const uint32_t waves = gridDim.x * (kBlock / 64);
    uint32_t c = (total / waves) & ~63u;
    return c < 64u ? 64u : (c > kRayChunk ? kRayChunk : c);
}
// Hands the idle lanes (mask idle_m) their next queue positions. Every lane of the wave must call it.
// Returns true for a lane that got one (in idx).
DEV bool feedLanes(WaveFeed& f, uint32_t* head, uint32_t total, uint32_t chunk, unsigned long long idle_m, bool idle,
                   uint32_t& idx) {
    const uint32_t lane = threadIdx.x & 63;
    if (f.next == f.end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, chunk);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= total) { f.drained = true; return false; }
        f.next = base;
        f.end = base + chunk < total ? base + chunk : total;
    }
    const uint32_t nidle = (uint32_t)__popcll(idle_m);
    const uint32_t avail = f.end - f.next;
    const uint32_t take = nidle < avail ? nidle : avail;
    const uint32_t rank = (uint32_t)__popcll(idle_m & ((1ULL << lane) - 1ULL));
    const bool got = idle && rank < take;
    if (got) idx = f.next + rank;
    f.next += take;
    return got;
}

// ---- trace: closest hits of the subpath rays of one depth ------------------------------------------------------
// Queue entry j of depth `depth` is subpath id q[j] (depth 0: id = j); its ray is the first line of its carry
// record. The closest hit (object -1: none) goes to the subpath's hit record.
// (Generating the subpaths inside the first launch of this kernel instead of in k_wf_gen was measured: the generation
// code costs the kernel a quarter of its occupancy, inlined or not, and the whole is slower: 411 against 418 Mpaths/s.)
#ifdef UBDPT_WAVE_TIMES
// measurement build: per-wave lifetime of k_wf_trace in ticks of the constant 100 MHz clock
// per depth: [0] waves that got work, [1] sum of lifetimes, [2] sum of time after the queue was found empty,
// [3] loop iterations, [4] iterations with a node step, [5] waves that found the queue empty at once,
// [6] longest lifetime, [7] ticks spent waiting for queue claims, [8] launches, [9] rays, [10] iterations with primitive tests
__device__ unsigned long long g_wave_times[16][12];
// phases of the loop, summed over waves (ticks): [0] arrivals, [1] feeding, [2] node fetch wait, [3] node step arithmetic,
// [4] primitive fetch wait, [5] primitive arithmetic
__device__ unsigned long long g_wave_phase[8];
// lanes by state at the node step, summed over iterations and waves: idle, waiting for their ray, at an inner node, at a leaf
__device__ unsigned long long g_wave_lanes[4];
#define WT_STAMP(k) { const unsigned long long n_ = wall_clock64(); wt_phase[k] += n_ - wt_last; wt_last = n_; }
#else
#define WT_STAMP(k)
#endif
template <bool STATS, int LDS_LEVEL>
__global__ __launch_bounds__(kBlock, traceWaves(LDS_LEVEL)) void k_wf_trace(RenderArgs A0, uint32_t depth) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ DObject s_objects[ldsObjects(LDS_LEVEL)];
    __shared__ LdsNode s_nodes[ldsNodes(LDS_LEVEL)];
    __shared__ Prim s_prims[ldsPrims(LDS_LEVEL)];
    RenderArgs A = A0;
    if (LDS_LEVEL >= 1) A.S.objects = stageObjects(A0.S, s_objects);
    if (LDS_LEVEL >= 2) stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    const uint64_t N = A.E.n;
    const uint64_t n2 = 2 * N;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = depth == 0 ? (uint32_t)n2 : A.wf_count[(uint64_t)depth * kWfQStride];
    const uint32_t* q = A.wf_trace_q + (uint64_t)(depth & 1) * n2;
    uint32_t* head = A.wf_count + (uint64_t)depth * kWfQStride + 3;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    TraceCounters* tc = STATS ? &ty.tc : nullptr;
    WaveFeed feed;
    feed.next = feed.end = 0; feed.drained = total == 0;
    const uint32_t chunk = chunkFor(total);
    int state = ST_IDLE;
    uint32_t idx = 0, id = 0, leaf_wait = 0;
    Ray ray;
    ray.o = ray.d = mk3(0, 0, 0);
    uint32_t cnt_word = 0;
    RayF rq = makeRayF(ray);
    float tmax = 0;
    int32_t node = kTraversalDone;
    Closest best;
    best.d = 0; best.u = 0; best.v = 0; best.object = -1; best.face = 0;
    LocalCache c;
    c.object = -1; c.gate_object = -1; c.gate_ok = false;
#ifdef UBDPT_WAVE_TIMES
    const unsigned long long wt_start = wall_clock64();
    unsigned long long wt_drained = 0, wt_iters = 0, wt_steps = 0, wt_claim = 0, wt_leaf = 0;
    unsigned long long wt_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_last = wt_start, wt_lanes[4] = {0, 0, 0, 0};
    if (gid == 0 && !STATS) { atomicAdd(&g_wave_times[depth][8], 1ULL); atomicAdd(&g_wave_times[depth][9], (unsigned long long)total); }
#endif
    for (;;) {
#ifdef UBDPT_WAVE_TIMES
        wt_iters++;
        if (feed.drained && !wt_drained) wt_drained = wall_clock64();
#endif
        // (1) lanes whose request of the previous iteration has arrived
        if (state == ST_RAY) {
            best.d = __builtin_inf(); best.u = 0; best.v = 0; best.object = -1; best.face = 0;
            if ((cnt_word & 0xffu) == 0) {  // no subpath under this id (light subpaths of a scene without emitters)
                state = ST_IDLE;
            } else if (S.num_prims == 0) {
                if (STATS) ty.closest++;
                storeHit(A, id, best);
                state = ST_IDLE;
            } else if (S.accel == 1) {
                if (STATS) ty.closest++;
                best = closestHitLinear(S, ray, tc);
                storeHit(A, id, best);
                state = ST_IDLE;
            } else {
                if (STATS) ty.closest++;
                rq = makeRayF(ray);
                tmax = floatUp(best.d);
                c.object = -1; c.gate_object = -1; c.gate_ok = false;
                st.sp = 0;
                node = 0;
                state = ST_ACTIVE;
            }
        }
        if (state == ST_ID) {
            const double* p = A.carry + (uint64_t)id * kCarryDoubles;
            ray.o = mk3(p[0], p[1], p[2]);
            ray.d = mk3(p[3], p[4], p[5]);
            cnt_word = reinterpret_cast<const uint32_t*>(p + kCarryCountSlot)[0];
            state = ST_RAY;
        }
        WT_STAMP(0)
        // (2) idle lanes take the next queue entries
        const unsigned long long idle_m = __ballot(state == ST_IDLE);
        const unsigned long long act_m = __ballot(state == ST_ACTIVE);
        if (!feed.drained && ((int)__popcll(idle_m) >= kRefillMin || act_m == 0)) {
#ifdef UBDPT_WAVE_TIMES
            const bool wt_c = feed.next == feed.end;
            const unsigned long long wt_t0 = wt_c ? wall_clock64() : 0;
#endif
            const bool fed = feedLanes(feed, head, total, chunk, idle_m, state == ST_IDLE, idx);
#ifdef UBDPT_WAVE_TIMES
            if (wt_c) wt_claim += wall_clock64() - wt_t0;
#endif
            if (fed) {
                if (depth == 0) {  // uniform
                    id = idx;
                    const double* p = A.carry + (uint64_t)id * kCarryDoubles;
                    ray.o = mk3(p[0], p[1], p[2]);
                    ray.d = mk3(p[3], p[4], p[5]);
                    cnt_word = reinterpret_cast<const uint32_t*>(p + kCarryCountSlot)[0];
                    state = ST_RAY;
                } else {
                    id = q[idx];
                    state = ST_ID;
                }
            }
        }
        WT_STAMP(1)
        if (feed.drained && idle_m == ~0ULL) break;
        if (act_m == 0) continue;
#ifdef UBDPT_WAVE_TIMES
        wt_steps++;
        wt_lanes[0] += __popcll(__ballot(state == ST_IDLE));
        wt_lanes[1] += __popcll(__ballot(state == ST_ID || state == ST_RAY));
        wt_lanes[2] += __popcll(__ballot(state == ST_ACTIVE && node >= 0));
        wt_lanes[3] += __popcll(__ballot(state == ST_ACTIVE && node < 0));
#endif
        // (3) one node step for every lane that stands at an inner node
#ifdef UBDPT_WAVE_TIMES
        BvhNode wt_nd;
        if (state == ST_ACTIVE && node >= 0) wt_nd = loadNode(S, node);
        __builtin_amdgcn_s_waitcnt(0);
        WT_STAMP(2)
        if (state == ST_ACTIVE && node >= 0) {
            node = descendNode(wt_nd, rq, tmax, st, tc);
            if (node == kTraversalDone) { storeHit(A, id, best); state = ST_IDLE; }
        }
        WT_STAMP(3)
#else
        if (state == ST_ACTIVE && node >= 0) {
            node = descendIn<(LDS_LEVEL >= 2)>(S, node, rq, tmax, st, tc);
            if (node == kTraversalDone) { storeHit(A, id, best); state = ST_IDLE; }
        }
#endif
        // (4) primitive tests, once enough lanes hold a leaf or nothing else can move
        const unsigned long long leaf_m = __ballot(state == ST_ACTIVE && node < 0);
        if (leaf_m != 0) {
            const unsigned long long inner_m = __ballot(state == ST_ACTIVE && node >= 0);
            if ((int)__popcll(leaf_m) >= kLeafBatch || inner_m == 0 || ++leaf_wait >= kLeafWait) {
                leaf_wait = 0;
#ifdef UBDPT_WAVE_TIMES
                wt_leaf++;
#endif
#ifdef UBDPT_WAVE_TIMES
                Prim wt_pr;
                if (state == ST_ACTIVE && node < 0) wt_pr = S.prims[~node];
                __builtin_amdgcn_s_waitcnt(0);
                WT_STAMP(4)
                const bool wt_on = state == ST_ACTIVE && node < 0;
                if (wt_on) cacheObject(S, ray, (int32_t)wt_pr.object, c);
                WT_STAMP(5)
                double wt_d = 0, wt_u = 0, wt_v = 0;
                bool wt_hit = false;
                if (wt_on) {
                    if (wt_pr.face == kSphereFace) wt_hit = intersectUnitSphere(c.l.o, c.l.d, wt_d);
                    else wt_hit = intersectTriangle(c.l.o, c.l.d, loadP(wt_pr.p), loadP(wt_pr.p + 3), loadP(wt_pr.p + 6), wt_d, wt_u, wt_v);
                }
                WT_STAMP(6)
                if (wt_on) {
                    if (wt_hit && lexLess(wt_d, (int32_t)wt_pr.object, wt_pr.face, best) &&
                        (wt_pr.face == kSphereFace || meshGate(S, (int32_t)wt_pr.object, c))) {
                        best.d = wt_d; best.u = wt_u; best.v = wt_v; best.object = (int32_t)wt_pr.object; best.face = wt_pr.face;
                    }
                }
                WT_STAMP(4)  // (measurement of the gate alone: its time is added to the primitive fetch column)
                if (wt_on) {
                    tmax = floatUp(best.d);
                    node = st.sp ? st.pop() : kTraversalDone;
                    if (node == kTraversalDone) { storeHit(A, id, best); state = ST_IDLE; }
                }
                WT_STAMP(7)
#else
                if (state == ST_ACTIVE && node < 0) {
                    if (STATS) { tc->prims++; countWaveStep(tc->wave_prims); }
                    testPrimClosest(S, ray, S.prims[~node], c, best);
                    tmax = floatUp(best.d);
                    node = st.sp ? st.pop() : kTraversalDone;
                    if (node == kTraversalDone) { storeHit(A, id, best); state = ST_IDLE; }
                }
#endif
            }
        }
    }
#ifdef UBDPT_WAVE_TIMES
    if ((threadIdx.x & 63) == 0 && !STATS) {
        const unsigned long long now = wall_clock64();
        unsigned long long* g = g_wave_times[depth];
        if (wt_iters > 2) {
            atomicAdd(&g[0], 1ULL);
            atomicAdd(&g[1], now - wt_start);
            atomicAdd(&g[2], now - (wt_drained ? wt_drained : now));
            atomicAdd(&g[3], wt_iters);
            atomicAdd(&g[4], wt_steps);
            atomicMax(&g[6], now - wt_start);
            atomicAdd(&g[7], wt_claim);
            atomicAdd(&g[10], wt_leaf);
            for (int k = 0; k < 8; k++) atomicAdd(&g_wave_phase[k], wt_phase[k]);
            for (int k = 0; k < 4; k++) atomicAdd(&g_wave_lanes[k], wt_lanes[k]);
        } else
            atomicAdd(&g[5], 1ULL);
    }
#endif
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// ---- shade ------------------------------------------------------------------------------------------------
// A workgroup takes 256 consecutive entries of the depth's trace queue: every lane looks at its subpath's hit
// record and draws the lobe (uapp/material.h), the chunk is ordered by (eye|light, lobe) with a stable counting
// sort in LDS (rank inside the wave by ballot, wave totals in LDS), and lane k then shades the k-th entry of the
// sorted chunk: the lanes of a wave run one BSDF. Misses and light subpaths that end at an absorbing hit
// (ubdptrenderer.cpp:460, :527) drop out in the sort.
// (Keeping the corner positions and shading attributes of the faces of a small scene in LDS as well removes the
// last dependent global fetch between a hit and its stores; measured: no difference, the kernel is bound by its
// stores. By their bytes, not their number: leaving out lines 1 and 2 of the vertex records takes a quarter off this
// kernel's time, while putting those lines through LDS, so that eight lanes write one record and the memory side sees
// requests of 64 bytes instead of 16, changes nothing. Three waves per SIMD instead of two spill 129 registers and
// are slower, 239 against 259 Mpaths/s.)
template <bool STATS, uint32_t LDS_OBJECTS>
__global__ __launch_bounds__(kBlock, UBDPT_SHADE_WAVES) void k_wf_shade(RenderArgs A0, uint32_t depth) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_wcnt[kWfRegions][kBlock / 64];
    __shared__ uint32_t s_sorted[kBlock];
    __shared__ DObject s_objects[LDS_OBJECTS ? LDS_OBJECTS : 1];
    RenderArgs A = A0;
    if (LDS_OBJECTS) A.S.objects = stageObjects(A0.S, s_objects);
    const DScene& S = A.S;
    const uint64_t N = A.E.n;
    const uint64_t n2 = 2 * N;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint64_t total = depth == 0 ? n2 : (uint64_t)A.wf_count[(uint64_t)depth * kWfQStride];
    const uint32_t* q = A.wf_trace_q + (uint64_t)(depth & 1) * n2;
    uint32_t* next_cnt = A.wf_count + (uint64_t)(depth + 1) * kWfQStride;
    uint32_t* next_q = A.wf_trace_q + (uint64_t)((depth + 1) & 1) * n2;
    Tally ty;
    for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < total; base += nthreads) {
        const uint64_t j = base + tid;
        int region = kWfRegions;  // dead
        uint32_t id = 0;
        if (j < total) {
            id = depth == 0 ? (uint32_t)j : q[j];
            const bool light = id >= N;
            const uint64_t i = light ? id - N : id;
            const PathStore& P = light ? A.L : A.E;
            const uint32_t m = P.meta(i);
            const uint32_t n = m & 0xffu;
            if (n > 0) {
                const unsigned long long w = (unsigned long long)__double_as_longlong(A.wf_hit[(uint64_t)id * kHitDoubles + 3]);
                const int32_t object = (int32_t)(uint32_t)w;
                if (object >= 0) {
                    uint64_t pixel;
                    Rng rng = pathRng(A, i, light, pixel);
                    const int lobe = pickLobe(S.objects[object], rng(hitDim(n, DIM_LOBE)));
                    if (!(light && lobe == LOBE_NONE)) region = (light ? 4 : 0) + lobe + 1;
                }
            }
            if (region == kWfRegions) (light ? A.fin_L : A.fin_E)[i] = m;  // the subpath ends here
        }
        // stable counting sort of the chunk by region
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
                uint32_t cw = s_wcnt[r][w];
                live += cw;
                if (r < region || (r == region && (uint32_t)w < wave)) slot += cw;
            }
        if (region < kWfRegions) s_sorted[slot] = id | ((uint32_t)region << 29);
        __syncthreads();
        bool cont = false;
        uint32_t sid = 0;
        if (tid < live) {
            const uint32_t e = s_sorted[tid];
            sid = e & 0x1fffffffu;
            const uint32_t r = e >> 29;
            const bool light = r >= 4;
            const int lobe = (int)(r & 3) - 1;
            const uint64_t i = light ? sid - N : sid;
            const PathStore& P = light ? A.L : A.E;
            uint32_t meta = P.meta(i);
            uint32_t n = meta & 0xffu;
            uint64_t pixel;
            Rng rng = pathRng(A, i, light, pixel);
            Carry cy;
            loadCarry(A, sid, cy, !light && n == 1);
            Closest hit;
            loadHit(A, sid, hit);
            cont = light ? shadeStep<STATS, true>(A, i, pixel, rng, n, meta, hit, lobe, cy, ty)
                         : shadeStep<STATS, false>(A, i, pixel, rng, n, meta, hit, lobe, cy, ty);
            if (cont) storeCarry(A, sid, cy);
            else (light ? A.fin_L : A.fin_E)[i] = meta;  // the subpath is complete
        }
        uint32_t pos = blockAppend(cont, &next_cnt[0], s_wave, &s_base);  // ends with a barrier: s_sorted / s_wcnt are free again
        if (cont) next_q[pos] = sid;
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
}

// ---- connections: the shadow rays of the pairs, visibility, then evaluation ---------------------------------
// One connection of pixel-sample i: everything renderPixel does for one (s,t) before it traces the shadow ray
// (ubdptrenderer.cpp:106-131) plus the geometric-normal test of connectionFactor (:622-626), which runs BEFORE the
// shadow ray here: it only removes shadow rays whose connection the reference would reject right after tracing them.
// The ray is the one UScene::visibility builds (uscene.cpp:21-27): from the light vertex towards the eye vertex.
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_pair_setup(RenderArgs A) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ v2d s_stage[kBlock / 64][64 * kStagePieces];
    const DScene& S = A.S;
    const uint32_t tid = threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    unsigned long long total = *A.pair_count;
    if (total > A.pair_capacity) total = A.pair_capacity;
    Tally ty;
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < total; base += nthreads) {
        const unsigned long long j = base + tid;
        bool ok = false;
        uint64_t p = 0;
        d3 lpos = mk3(0, 0, 0), epos = mk3(0, 0, 0);
        if (j < total) {
            p = A.pairs[j];
            const uint32_t s = (uint32_t)((p >> 8) & 0xff), t = (uint32_t)(p & 0xff);
            const uint64_t i = p >> 16;
            if (s != 0) {
                const double* rl = A.L.rp(s - 1, i);
                const double* re = A.E.rp(t - 1, i);
                lpos = rld3(rl, F_POS);
                epos = rld3(re, F_POS);
                ok = true;
                if (t == 1) {
                    uint32_t tx, tyy;
                    ok = t1Pixel(S, lpos, epos, tx, tyy);
                }
                if (ok) {
                    d3 ce = lpos - epos;
                    ce = ce / sqrt(dot(ce, ce));
                    if ((dot(ce, t == 1 ? lensNormal(S) : rld3(re, F_NG)) <= 0) || (dot(-ce, rld3(rl, F_NG)) <= 0)) ok = false;
                }
            }
        }
        const uint32_t pos = blockAppend(ok, A.ray_count, s_wave, &s_base);
        // rays never outnumber pairs, and the ray queue holds as many entries as the pair queue. The rays of a wave's
        // lanes go to consecutive places of the queue: written together, as whole lines (waveStoreRanked)
        const double dist = length(epos - lpos);
        const d3 dir = (epos - lpos) / dist;
        const unsigned long long okm = __ballot(ok);
        if (okm) {
            const int lane = (int)(tid & 63);
            const uint32_t first = __shfl(pos, __ffsll((long long)okm) - 1, 64);   // place of the wave's first ray
            v2d pc[4] = {mk2(lpos.x, lpos.y), mk2(lpos.z, dir.x), mk2(dir.y, dir.z), mk2(dist, 0.0)};
            waveStoreRanked<4>(s_stage[tid >> 6], lane, ok, pc, A.rays + (uint64_t)first * kRayDoubles);
            if (ok) A.ray_pairs[pos] = p;
        }
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
}

// UScene::visibility (uscene.cpp:28-38) for every ray of the queue: flag 1 when nothing blocks it.
template <bool STATS, int LDS_LEVEL>
__global__ __launch_bounds__(kBlock, traceWaves(LDS_LEVEL)) void k_rays_visible(RenderArgs A0) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ DObject s_objects[ldsObjects(LDS_LEVEL)];
    __shared__ LdsNode s_nodes[ldsNodes(LDS_LEVEL)];
    __shared__ Prim s_prims[ldsPrims(LDS_LEVEL)];
    RenderArgs A = A0;
    if (LDS_LEVEL >= 1) A.S.objects = stageObjects(A0.S, s_objects);
    if (LDS_LEVEL >= 2) stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long total64 = *A.ray_count;
    if (total64 > A.pair_capacity) total64 = A.pair_capacity;
    const uint32_t total = (uint32_t)total64;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    TraceCounters* tc = STATS ? &ty.tc : nullptr;
    WaveFeed feed;
    feed.next = feed.end = 0; feed.drained = total == 0;
    const uint32_t chunk = chunkFor(total);
    int state = ST_IDLE;
    uint32_t idx = 0, leaf_wait = 0;
    Ray ray;
    ray.o = ray.d = mk3(0, 0, 0);
    double dist = 0;
    RayF rq = makeRayF(ray);
    float tmax = 0;
    int32_t node = kTraversalDone;
    LocalCache c;
    c.object = -1; c.gate_object = -1; c.gate_ok = false;
    for (;;) {
        // (1) rays requested in the previous iteration
        if (state == ST_RAY) {
            if (STATS) ty.shadow++;
            if (S.num_prims == 0) {
                A.ray_flag[idx] = 1;
                state = ST_IDLE;
            } else if (S.accel == 1) {
                A.ray_flag[idx] = visibleLinearRay(S, ray, dist, tc) ? 1 : 0;
                state = ST_IDLE;
            } else {
                // The tree is walked from the END of the segment (the eye vertex, which the neighbouring rays of the
                // queue share): which boxes a segment meets does not depend on the direction it is walked in, and
                // the primitive tests below use the ray as the reference builds it.
                Ray back;
                back.o = ray.o + ray.d * dist;
                back.d = -ray.d;
                rq = makeRayF(back);   // (2 M triangles: this kernel 7.36 -> 7.19 ms per batch against walking from the light vertex)
                tmax = floatUp(dist);
                c.object = -1; c.gate_object = -1; c.gate_ok = false;
                st.sp = 0;
                node = 0;
                state = ST_ACTIVE;
            }
        }
        // (2) idle lanes take the next rays
        const unsigned long long idle_m = __ballot(state == ST_IDLE);
        const unsigned long long act_m = __ballot(state == ST_ACTIVE);
        if (!feed.drained && ((int)__popcll(idle_m) >= kRefillMin || act_m == 0)) {
            if (feedLanes(feed, A.ray_head, total, chunk, idle_m, state == ST_IDLE, idx)) {
                const double* p = A.rays + (uint64_t)idx * kRayDoubles;
                ray.o = mk3(p[0], p[1], p[2]);
                ray.d = mk3(p[3], p[4], p[5]);
                dist = p[6];
                state = ST_RAY;
            }
        }
        if (feed.drained && idle_m == ~0ULL) break;
        if (act_m == 0) continue;
        // (3) node steps
        if (state == ST_ACTIVE && node >= 0) {
            node = descendIn<(LDS_LEVEL >= 2)>(S, node, rq, tmax, st, tc);
            if (node == kTraversalDone) { A.ray_flag[idx] = 1; state = ST_IDLE; }
        }
        // (4) primitive tests
        const unsigned long long leaf_m = __ballot(state == ST_ACTIVE && node < 0);
        if (leaf_m != 0) {
            const unsigned long long inner_m = __ballot(state == ST_ACTIVE && node >= 0);
            if ((int)__popcll(leaf_m) >= kLeafBatch || inner_m == 0 || ++leaf_wait >= kLeafWait) {
                leaf_wait = 0;
                if (state == ST_ACTIVE && node < 0) {
                    if (STATS) { tc->prims++; countWaveStep(tc->wave_prims); }
                    if (testPrimBlocks(S, ray, S.prims[~node], c, dist)) { A.ray_flag[idx] = 0; state = ST_IDLE; }
                    else {
                        node = st.sp ? st.pop() : kTraversalDone;
                        if (node == kTraversalDone) { A.ray_flag[idx] = 1; state = ST_IDLE; }
                    }
                }
            }
        }
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// ---- scenes whose tree and primitives sit in LDS (a few dozen primitives): one ray per lane ------------------------
// A ray crosses such a tree in two or three steps out of LDS and the rays of a wave take about equally long, so the
// queue of the persistent waves and the ray records between set-up and traversal cost more than they save (measured on
// a box of 14 primitives at 512 x 512: trace 0.64 -> 0.43 ms per launch, connection visibility 6.0 -> 1.9 ms per batch).
template <bool STATS>
__global__ __launch_bounds__(kBlock, UBDPT_TRACE_WAVES) void k_wf_trace_lds(RenderArgs A0, uint32_t depth) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ DObject s_objects[ldsObjects(2)];
    __shared__ LdsNode s_nodes[ldsNodes(2)];
    __shared__ Prim s_prims[ldsPrims(2)];
    RenderArgs A = A0;
    A.S.objects = stageObjects(A0.S, s_objects);
    stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    const uint64_t n2 = 2 * A.E.n;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint64_t total = depth == 0 ? n2 : (uint64_t)A.wf_count[(uint64_t)depth * kWfQStride];
    const uint32_t* q = A.wf_trace_q + (uint64_t)(depth & 1) * n2;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    for (uint64_t j = gid; j < total; j += nthreads) {
        const uint32_t id = depth == 0 ? (uint32_t)j : q[j];
        const double* p = A.carry + (uint64_t)id * kCarryDoubles;
        if ((reinterpret_cast<const uint32_t*>(p + kCarryCountSlot)[0] & 0xffu) == 0) continue;  // no subpath under this id
        Ray ray;
        ray.o = mk3(p[0], p[1], p[2]);
        ray.d = mk3(p[3], p[4], p[5]);
        storeHit(A, id, closestHit<STATS, true>(S, ray, st, ty));
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// Set-up tests, shadow ray and compaction of the visible pairs in one kernel (k_pair_setup + k_rays_visible +
// k_pair_compact of the large scenes).
template <bool STATS>
__global__ __launch_bounds__(kBlock, UBDPT_TRACE_WAVES) void k_pair_visible_lds(RenderArgs A0) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ DObject s_objects[ldsObjects(2)];
    __shared__ LdsNode s_nodes[ldsNodes(2)];
    __shared__ Prim s_prims[ldsPrims(2)];
    RenderArgs A = A0;
    A.S.objects = stageObjects(A0.S, s_objects);
    stageTree(A.S, s_nodes, s_prims);
    const DScene& S = A.S;
    const uint32_t tid = threadIdx.x;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + tid;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    unsigned long long total = *A.pair_count;
    if (total > A.pair_capacity) total = A.pair_capacity;
    Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid, A.ovf_cap);
    Tally ty;
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < total; base += nthreads) {
        const unsigned long long j = base + tid;
        bool ok = false;
        uint64_t p = 0;
        if (j < total) {
            p = A.pairs[j];
            const uint32_t s = (uint32_t)((p >> 8) & 0xff), t = (uint32_t)(p & 0xff);
            const uint64_t i = p >> 16;
            if (s != 0) {
                const double* rl = A.L.rp(s - 1, i);
                const double* re = A.E.rp(t - 1, i);
                const d3 lpos = rld3(rl, F_POS);
                const d3 epos = rld3(re, F_POS);
                ok = true;
                if (t == 1) {
                    uint32_t tx, tyy;
                    ok = t1Pixel(S, lpos, epos, tx, tyy);
                }
                if (ok) {
                    d3 ce = lpos - epos;
                    ce = ce / sqrt(dot(ce, ce));
                    if ((dot(ce, t == 1 ? lensNormal(S) : rld3(re, F_NG)) <= 0) || (dot(-ce, rld3(rl, F_NG)) <= 0)) ok = false;
                }
                if (ok) ok = visible<STATS, true>(S, lpos, epos, st, ty);
            }
        }
        const uint32_t pos = blockAppend(ok, A.vis_count, s_wave, &s_base);
        if (ok) A.vis_pairs[pos] = p;
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
    if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
}

// The pairs whose shadow ray found nothing in the way, in queue order. A lane looks at four consecutive rays (one
// word of flags), so a workgroup appends the survivors of 1024 rays with one atomic: appends to one counter cost a
// few nanoseconds each whatever their size, and there are tens of millions of rays.
__global__ __launch_bounds__(kBlock) void k_pair_compact(RenderArgs A) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    unsigned long long total = *A.ray_count;
    if (total > A.pair_capacity) total = A.pair_capacity;
    const unsigned long long words = (total + 3) / 4;
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(A.ray_flag);
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < words; base += nthreads) {
        const unsigned long long w = base + tid;
        uint32_t f = 0;
        if (w < words) {
            f = flags[w];
            const unsigned long long left = total - 4 * w;  // rays this word covers (the last word may be partial)
            if (left < 4) f &= (1u << (8 * (uint32_t)left)) - 1u;
        }
        const uint32_t n = (f & 1u) + ((f >> 8) & 1u) + ((f >> 16) & 1u) + ((f >> 24) & 1u);
        // exclusive prefix of n over the workgroup
        uint32_t incl = waveInclusiveScan(n);
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int k = 0; k < kBlock / 64; k++) tot += s_wave[k];
            s_base = tot ? atomicAdd(A.vis_count, tot) : 0u;
        }
        __syncthreads();
        uint32_t pos = s_base + incl - n;
        for (uint32_t k = 0; k < (tid >> 6); k++) pos += s_wave[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((f >> (8 * k)) & 1u) A.vis_pairs[pos++] = A.ray_pairs[4 * w + k];
    }
}

// Adjacent lanes mostly hold pairs of one pixel-sample that land in the same pixel (the queue is in
// pixel-sample order, k_pair_enum emits the own-pixel pairs of a pixel-sample next to each other):
// runs of equal target pixels are summed with shuffles and the first lane of a run issues the three
// atomics. What limits this kernel is the gather of the two end-vertex records (24 loads of 16 bytes per lane, every
// lane another line): with plain adds in place of the atomics it takes the same time, with both records read from
// one address per wave a quarter less (2 M-triangle scene: 2.5 -> 1.9 ms, 263 -> 279 Mpaths/s, wrong image).
// Pairs whose s or t lies beyond the dense vertex slots are put on a list (A.ray_pairs, free again by now; its length is
// A.ray_count[3]) instead of being evaluated here: k_pair_overflow, launched behind this kernel, evaluates them with the
// page lookups in its record addresses, and this kernel's addresses stay plain arithmetic.
template <bool STATS>
__global__ __launch_bounds__(kBlock, UBDPT_EVAL_WAVES) void k_pair_eval(RenderArgs A) {
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t total = (uint64_t)*A.vis_count;
    const uint64_t* queue = A.vis_pairs;
    const uint64_t npix = (uint64_t)A.S.img_res_x * A.S.img_res_y;
    const uint32_t dense = A.E.dense;  // both stores have the same number of dense slots
    Tally ty;
    for (uint64_t base = gid - lane; base < total; base += nthreads) {  // wave-uniform trip count
        const uint64_t j = base + lane;
        uint64_t key = ~0ULL;  // target pixel, ~0: nothing to add
        d3 v = mk3(0, 0, 0);
        if (j < total) {
            uint64_t p = queue[j];
            const uint32_t s = (uint32_t)((p >> 8) & 0xff), t = (uint32_t)(p & 0xff);
            uint64_t pixel;
            if (s > dense || t > dense) {
                A.ray_pairs[atomicAdd(&A.ray_count[3], 1u)] = p;
            } else if (pairEval<STATS, false>(A, p >> 16, s, t, ty, pixel, v)) key = pixel;
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
