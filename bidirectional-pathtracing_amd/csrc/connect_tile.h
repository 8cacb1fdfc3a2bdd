// connect_tile.h — the (s,t) connection stage as an LDS-tiled kernel (included by render.hip).
//
// renderPixel's connection loop (uengine/ubdptrenderer.cpp:103-149) reads every vertex of both
// subpaths many times: vertex s of the light subpath for every t, vertex t of the eye subpath for
// every s, and the pdf chains of all earlier vertices for every MIS weight. Gathering those from the
// [field][slot][path] store per pair fetches every vertex ~2.7 times and is bound by the number of
// L1 misses the memory pipeline can keep in flight. Here a workgroup owns kTilePaths consecutive
// pixel-samples and works in phases whose live state is small:
//   A. stage    the vertices of their subpaths are copied ONCE from HBM into LDS, with loads that run
//               along the path index (contiguous in the store), as records of kVS doubles;
//   B. count    connectable vertices per subpath, prefix sum of the pair counts (one wave);
//   C. visible  one pair per lane: pixel projection (t = 1), geometric-normal test, shadow ray;
//               the surviving pairs are compacted into an LDS list (ballot + wave totals);
//   D. evaluate one visible pair per lane, entirely from LDS: connection factor, MIS weight;
//               contributions to the pixel-sample's own pixel (t >= 2) are summed in LDS;
//   E. flush    three atomics per pixel-sample into the sum buffer; t = 1 splats went there directly.
// Pixel-samples whose vertices do not fit the tile (very long subpaths) take the per-pair HBM path.
#pragma once

namespace ub {

constexpr int kTilePaths = 32;    // pixel-samples per workgroup
constexpr int kTileVerts = 288;   // staged vertices per workgroup (mean is ~6.5 per pixel-sample)
constexpr int kTilePairs = 768;   // pairs per round of phases C and D
constexpr int kVS = 25;           // doubles per staged vertex: 22 fields, packed (flags, object), 2 spare; odd stride
constexpr int kVInts = 22;        //   spreads a field of consecutive vertices over the LDS banks

struct LdsSubpath {
    const double* base;  // staged vertex 0 of the subpath
    DEV double ld(int f, uint32_t slot, uint64_t) const { return base[slot * kVS + f]; }
    DEV d3 ld3(int f, uint32_t slot) const {
        const double* p = base + slot * kVS + f;
        return mk3(p[0], p[1], p[2]);
    }
    DEV unsigned long long ints(uint32_t slot) const {
        return (unsigned long long)__double_as_longlong(base[slot * kVS + kVInts]);
    }
    DEV bool specular(uint32_t slot, uint64_t) const { return (ints(slot) >> 8) & 1; }
};

DEV void loadConnLds(const LdsSubpath& P, uint32_t slot, ConnVertex& c) {
    c.pos = P.ld3(F_POS, slot); c.Ng = P.ld3(F_NG, slot); c.Ns = P.ld3(F_NS, slot);
    c.Ts = P.ld3(F_TS, slot);   c.Bs = P.ld3(F_BS, slot); c.a = P.ld3(F_A, slot);
    c.p_light_A = P.ld(F_PL, slot, 0); c.p_eye_A = P.ld(F_PE, slot, 0);
    c.tex_u = P.ld(F_U, slot, 0); c.tex_v = P.ld(F_V, slot, 0);
    unsigned long long w = P.ints(slot);
    int f = (int)(w & 0xffffffffULL);
    c.object = (int32_t)(w >> 32);
    c.lobe = (f & 0xff) - 1;
    c.specular = (f >> 8) & 1;
}

// index of the k-th (0-based) set bit of m
DEV uint32_t nthSetBit(uint32_t m, uint32_t k) {
    for (uint32_t q = 0; q < k; q++) m &= m - 1;
    return (uint32_t)__ffs((int)m) - 1;
}

template <bool STATS>
__global__ __launch_bounds__(kBlock, 2) void k_connect_tile(RenderArgs A) {  // LDS allows two tiles per CU
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    __shared__ double s_v[kTileVerts * kVS];
    __shared__ double s_I[kTilePaths * 3];
    __shared__ uint32_t s_vis[kTilePairs];   // visible pairs of the round: p | s << 8 | t << 16
    __shared__ uint32_t s_nE[kTilePaths], s_nL[kTilePaths], s_ofs[kTilePaths], s_staged[kTilePaths];
    __shared__ uint32_t s_okE[kTilePaths], s_okL[kTilePaths], s_pre[kTilePaths + 1], s_max[2];
    __shared__ uint32_t s_wave[kBlock / 64], s_nvis;
    const DScene& S = A.S;
    const uint32_t tid = threadIdx.x;
    const uint64_t N = A.E.n;
    const uint64_t npix = (uint64_t)S.img_res_x * S.img_res_y;
    const uint64_t path0 = (uint64_t)blockIdx.x * kTilePaths;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + tid;
    Tally ty;

    // ---- A0. subpath lengths and the place of every pixel-sample's vertices in the tile (wave 0)
    if (tid < 64) {
        uint32_t p = tid;
        uint64_t i = path0 + p;
        uint32_t nE = 0, nL = 0;
        if (p < kTilePaths && i < N) { nE = A.E.count[i]; nL = A.L.count[i]; }
        uint32_t nv = nE + nL;
        bool want = nv > 0 && nE <= 32 && nL <= 32;
        uint32_t incl = waveInclusiveScan(want ? nv : 0u);
        bool staged = want && incl <= (uint32_t)kTileVerts;
        uint32_t mE = waveMax(staged ? nE : 0u), mL = waveMax(staged ? nL : 0u);
        if (p < kTilePaths) {
            s_nE[p] = nE; s_nL[p] = nL; s_ofs[p] = incl - (want ? nv : 0u); s_staged[p] = staged ? 1u : 0u;
            s_I[3 * p] = 0; s_I[3 * p + 1] = 0; s_I[3 * p + 2] = 0;
        }
        if (tid == 0) { s_max[0] = mE; s_max[1] = mL; }
        if (STATS && p < kTilePaths && i < N) { Tally t0; flushTally(t0, A.counters, nE, nL); }
    }
    __syncthreads();

    // ---- A. stage: a row is (slot, field) of one subpath kind; 32 consecutive lanes read 32 consecutive paths
    {
        const uint32_t p = tid & (kTilePaths - 1);
        const uint32_t r0 = tid / kTilePaths;
        const uint64_t i = path0 + p;
        const bool staged = s_staged[p] != 0;
        const uint32_t nE = s_nE[p], nL = s_nL[p], ofs = s_ofs[p];
        for (int sub = 0; sub < 2; sub++) {
            const PathStore& P = sub ? A.L : A.E;
            const uint32_t n = sub ? nL : nE;
            const uint32_t vbase = ofs + (sub ? nE : 0u);
            const uint32_t rows = s_max[sub] * (NUM_F + 1);
            constexpr uint32_t kRowsPerStep = kBlock / kTilePaths;  // 8 rows per workgroup step
            constexpr int kInFlight = 8;                            // loads a lane keeps in flight
            for (uint32_t row0 = r0; row0 < rows; row0 += kRowsPerStep * kInFlight) {
                double vals[kInFlight];
#pragma unroll
                for (int u = 0; u < kInFlight; u++) {
                    uint32_t row = row0 + kRowsPerStep * u;
                    uint32_t k = row / (NUM_F + 1), f = row % (NUM_F + 1);
                    vals[u] = 0;
                    if (staged && row < rows && k < n) {
                        if (f < (uint32_t)NUM_F) vals[u] = P.v[P.at((int)f, k, i)];
                        else {
                            unsigned long long w =
                                ((unsigned long long)(uint32_t)P.obj(k, i) << 32) | (uint32_t)P.flg(k, i);
                            vals[u] = __longlong_as_double((long long)w);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kInFlight; u++) {
                    uint32_t row = row0 + kRowsPerStep * u;
                    uint32_t k = row / (NUM_F + 1), f = row % (NUM_F + 1);
                    if (staged && row < rows && k < n) s_v[(vbase + k) * kVS + f] = vals[u];
                }
            }
        }
    }
    __syncthreads();

    // ---- B. connectable vertices (exist, not specular) and the pair prefix sum (wave 0)
    if (tid < 64) {
        uint32_t p = tid;
        uint32_t okE = 0, okL = 0, cnt = 0;
        if (p < kTilePaths && s_staged[p]) {
            LdsSubpath Ev{&s_v[s_ofs[p] * kVS]}, Lv{&s_v[(s_ofs[p] + s_nE[p]) * kVS]};
            for (uint32_t t = 0; t < s_nE[p]; t++) okE |= Ev.specular(t, 0) ? 0u : (1u << t);
            for (uint32_t s = 0; s < s_nL[p]; s++) okL |= Lv.specular(s, 0) ? 0u : (1u << s);
            cnt = (uint32_t)__popc(okE) * (uint32_t)__popc(okL);
        }
        uint32_t incl = waveInclusiveScan(cnt);
        if (p < kTilePaths) { s_okE[p] = okE; s_okL[p] = okL; s_pre[p + 1] = incl; }
        if (tid == 0) s_pre[0] = 0;
    }
    __syncthreads();

    const uint32_t total = s_pre[kTilePaths];
    for (uint32_t round0 = 0; round0 < total; round0 += kTilePairs) {
        if (tid == 0) s_nvis = 0;
        __syncthreads();
        // ---- C. visibility, survivors compacted into s_vis
        {
            Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid);
            for (uint32_t it = 0; it < kTilePairs / kBlock; it++) {
                uint32_t j = round0 + it * kBlock + tid;
                bool vis = false;
                uint32_t packed = 0;
                if (j < total) {
                    uint32_t p = 0;
#pragma unroll
                    for (int step = kTilePaths / 2; step >= 1; step >>= 1)
                        if (s_pre[p + step] <= j) p += step;
                    uint32_t local = j - s_pre[p];
                    uint32_t okE = s_okE[p], okL = s_okL[p];
                    uint32_t cE = (uint32_t)__popc(okE);
                    uint32_t s = nthSetBit(okL, local / cE) + 1, t = nthSetBit(okE, local % cE) + 1;
                    packed = p | (s << 8) | (t << 16);
                    LdsSubpath Ev{&s_v[s_ofs[p] * kVS]}, Lv{&s_v[(s_ofs[p] + s_nE[p]) * kVS]};
                    d3 lpos = Lv.ld3(F_POS, s - 1), epos = Ev.ld3(F_POS, t - 1);
                    bool ok = true;
                    if (t == 1) {
                        uint32_t tx, tyy;
                        ok = t1Pixel(S, lpos, epos, tx, tyy);
                    }
                    if (ok) {  // geometric-normal test of connectionFactor (:622-626), ahead of the shadow ray
                        d3 ce = lpos - epos;
                        ce = ce / sqrt(dot(ce, ce));
                        ok = !((dot(ce, Ev.ld3(F_NG, t - 1)) <= 0) || (dot(-ce, Lv.ld3(F_NG, s - 1)) <= 0));
                    }
                    if (ok) vis = visible<STATS>(S, lpos, epos, st, ty);
                }
                unsigned long long m = __ballot(vis);
                if ((tid & 63) == 0) s_wave[tid >> 6] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t pos = s_nvis + (uint32_t)__popcll(m & ((1ULL << (tid & 63)) - 1ULL));
                for (uint32_t w = 0; w < (tid >> 6); w++) pos += s_wave[w];
                if (vis) s_vis[pos] = packed;
                __syncthreads();
                if (tid == 0) s_nvis += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
                __syncthreads();
            }
            if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
        }
        // ---- D. evaluation of the visible pairs, from LDS
        const uint32_t nvis = s_nvis;
        for (uint32_t k = tid; k < nvis; k += kBlock) {
            uint32_t packed = s_vis[k];
            uint32_t p = packed & 0xff, s = (packed >> 8) & 0xff, t = packed >> 16;
            LdsSubpath Ev{&s_v[s_ofs[p] * kVS]}, Lv{&s_v[(s_ofs[p] + s_nE[p]) * kVS]};
            ConnVertex vl, ve;
            loadConnLds(Lv, s - 1, vl);
            loadConnLds(Ev, t - 1, ve);
            d3 lprev = s > 1 ? Lv.ld3(F_POS, s - 2) : mk3(0, 0, 0);
            d3 eprev = t > 1 ? Ev.ld3(F_POS, t - 2) : mk3(0, 0, 0);
            d3 c;
            double w;
            if ((s == 1 || vl.lobe == LOBE_LAMBERTIAN) && (t == 1 || ve.lobe == LOBE_LAMBERTIAN)) {
                if (!connectLambert(S, Lv, Ev, 0, vl, ve, s, t, lprev, eprev, c, w)) continue;
            } else {
                if (!connectionFactorTail(S, vl, ve, s, t, lprev, eprev, c)) continue;
                w = misWeight(S, Lv, Ev, 0, vl, ve, s, t, lprev, eprev);
            }
            d3 val = vl.a * ve.a * c * w;
            if (STATS) ty.conn++;
            if (t == 1) {
                uint32_t tx = 0, tyy = 0;
                if (!t1Pixel(S, vl.pos, ve.pos, tx, tyy)) continue;
                if (STATS) { ty.splat++; ty.fb++; }
                fbAdd(A.fb, npix, (uint64_t)tyy * S.img_res_x + tx, val);
            } else {
                atomicAdd(&s_I[3 * p], val.x);
                atomicAdd(&s_I[3 * p + 1], val.y);
                atomicAdd(&s_I[3 * p + 2], val.z);
            }
        }
        __syncthreads();
    }

    // ---- pixel-samples that did not fit the tile: per-pair evaluation straight from HBM
    if (tid < kTilePaths && !s_staged[tid] && (s_nE[tid] + s_nL[tid]) > 0) {
        Stack st = makeStack(lds_stack, A.ovf, A.ovf_stride, gid);
        uint64_t i = path0 + tid;
        for (uint32_t s = 1; s <= s_nL[tid]; s++) {
            if (A.L.specular(s - 1, i)) continue;
            for (uint32_t t = 1; t <= s_nE[tid]; t++) {
                if (A.E.specular(t - 1, i)) continue;
                connectPair<STATS>(A, i, s, t, st, ty);
            }
        }
        if (st.overflowed) atomicAdd(&A.counters[C_STACK_OVERFLOW], 1ULL);
    }
    __syncthreads();

    // ---- E. own-pixel sums of the tile -> sum buffer
    if (tid < kTilePaths * 3) {
        uint32_t p = tid / 3, ch = tid % 3;
        double v = s_I[tid];
        uint64_t i = path0 + p;
        if (v != 0 && i < N) {
            unsafeAtomicAdd(A.fb + (uint64_t)ch * npix + (i % npix), v);
            if (STATS && ch == 0) ty.fb++;
        }
    }
    if (STATS) flushTally(ty, A.counters, 0, 0);
}

}  // namespace ub
