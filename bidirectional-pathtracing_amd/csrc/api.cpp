// api.cpp — the C ABI of include/ubdpt.h: scene flattening, device residency, pass scheduling.
//
// Host-side counterpart of UBDPTRenderer::initialize / renderPass / stop
// (uengine/ubdptrenderer.cpp:7-86) and of the constructors that derive per-object data in the
// reference (uapp/object.cpp:5-12, uapp/emitter.cpp:3-7, uapp/mesh.cpp:3-106,
// uengine/uscene.cpp:4-19, uengine/ucamera.cpp:3-14,67-70). Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "render.h"

using namespace ub;

namespace {

double nowSeconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// uengine/ugeometry.cpp:4-15
double triangleArea(d3 p0, d3 p1, d3 p2) {
    d3 AB = p1 - p0;
    d3 AC = p2 - p0;
    double l1 = length(AB);
    double l2 = length(AC);
    double c = dot(AB / l1, AC / l2);
    double s = std::sqrt(1 - c * c);
    return 0.5 * l1 * l2 * s;
}

d3 vpos(const ubdpt_mesh_vertex& v) { return mk3((double)v.pos[0], (double)v.pos[1], (double)v.pos[2]); }

// uapp/mesh.cpp:38-86
m4 boundingSphere(const ubdpt_mesh_vertex* verts, uint64_t first, uint64_t n) {
    double max_d = 0;
    d3 x = vpos(verts[3 * first]);
    d3 y = mk3(0, 0, 0), z = mk3(0, 0, 0);
    for (uint64_t i = 3 * first; i < 3 * (first + n); i++) {
        double d = distance(vpos(verts[i]), x);
        if (d > max_d) { max_d = d; y = vpos(verts[i]); }
    }
    max_d = 0;
    for (uint64_t i = 3 * first; i < 3 * (first + n); i++) {
        double d = distance(vpos(verts[i]), y);
        if (d > max_d) { max_d = d; z = vpos(verts[i]); }
    }
    d3 center = 0.5 * (y + z);
    double radius = 0.5 * max_d;
    for (uint64_t i = 3 * first; i < 3 * (first + n); i++) {
        double d = distance(vpos(verts[i]), center);
        if (d > radius) radius = d;
    }
    return inverse(scale(translate(identity4(), center), mk3(radius, radius, radius)));
}

enum KernelId { K_EYE = 0, K_LIGHT, K_CONNECT, K_PAIR_ENUM, K_PAIR_SETUP, K_RAYS_VISIBLE, K_PAIR_COMPACT,
                K_PAIR_EVAL, K_WF_GEN, K_WF_TRACE, K_WF_SHADE, K_WF_TAIL, K_INTERSECT, K_VISIBILITY, K_NUM };
const char* kKernelNames[K_NUM] = {"k_eye", "k_light", "k_connect", "k_pair_enum", "k_pair_setup", "k_rays_visible",
                                   "k_pair_compact", "k_pair_eval", "k_wf_gen", "k_wf_trace",
                                   "k_wf_shade", "k_wf_tail", "k_intersect_batch", "k_visibility_batch"};
constexpr int kLastRenderKernel = K_WF_TAIL;

struct TimedLaunch {
    hipEvent_t start, stop;
    int kernel;
    double units;
};

template <class T>
hipError_t upload(T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice);
    return e;
}

}  // namespace

struct ubdpt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    ubdpt_options opt{};
    ubdpt_render_params params{};

    std::vector<DObject> h_objects;
    DObject* d_objects = nullptr;
    int32_t* d_emitters = nullptr;
    float* d_face_pos = nullptr;
    FaceAttr* d_face_attr = nullptr;
    double* d_face_prob = nullptr;
    double* d_face_cdf = nullptr;
    DTexture* d_textures = nullptr;
    uint8_t* d_texels = nullptr;
    BvhNode* d_nodes = nullptr;
    ExactNode* d_xnodes = nullptr;
    Prim* d_prims = nullptr;
    DScene S{};

    uint64_t npix = 0;
    double* d_fb = nullptr;       // planar; the sum buffer, followed by pass_slots per-pass sum buffers
    uint32_t pass_slots = 0;      // > 0: ubdpt_render_passes_separate keeps up to that many passes apart
    // ubdpt_enqueue_passes_separate: a fence is one event per lane, recorded behind the batches of a call; a per-pass
    // buffer remembers the fence of the call that filled it last, and reading it waits for that fence only, not for
    // passes enqueued later into other buffers. -1: filled by a call that waited itself.
    static constexpr int kFences = 4;
    std::vector<hipEvent_t> fences[kFences];
    uint32_t next_fence = 0;
    int slot_fence[64];
    hipEvent_t clear_ev = nullptr;  // the per-pass buffers of a call have been cleared (the lanes wait for it)
    double* d_staging = nullptr;  // interleaved, npix * 3
    double* h_staging = nullptr;  // pinned host mirror of d_staging, allocated on first use (read-backs that add into the caller's buffer)
    uint64_t cap_paths = 0;       // path slots allocated per batch lane
    uint32_t dense_slots = 0;     // vertex slots stored for every subpath (the rest come from a pool of pages)
    uint32_t pages_per_k = 0;     // pool size: pages per 1024 subpaths of a batch
    uint32_t passes_in_flight = 1;
    uint32_t wf_depth = 8;        // bounces run as wavefront launches before the tail loop
    uint64_t ovf_threads = 0;     // path slots of a lane, rounded up to whole workgroups
    uint64_t ovf_lanes = 0;       // lanes the overflow-stack buffer is allocated for (= its stride)
    uint32_t ovf_cap = kLdsStack + kOverflowStack;  // entries of a lane's whole traversal stack (on-chip part included)
    void* noalloc_scratch = nullptr;             // device address of the static scratch array on this context's GPU
    unsigned long long* h_noalloc_counters = nullptr;  // pinned, kNumCounters words: counters of the last no-allocation call
    unsigned long long pair_capacity = 0;
    // One call's passes as the batches see them, and one batch: pixel-samples [pos, pos + n) of the call in (pass, pixel) order
    struct CallDesc { uint64_t first_pass = 0, pass_stride = 1, seed = 0; bool separate = false; uint32_t first_slot = 0; };
    struct BatchDesc { CallDesc call; uint64_t pos = 0, n = 0; };
    // A batch lane owns every per-batch buffer and a stream; consecutive batches go to consecutive
    // lanes, so the narrow late stages of one batch (deep bounces, tail) overlap the wide early
    // stages of the next. The sum buffer and the counters are shared (atomics).
    struct BatchLane {
        hipStream_t stream = nullptr;
        PathStore E{}, L{};
        int32_t* d_ovf = nullptr;
        uint64_t* d_pairs = nullptr;
        unsigned long long* d_pair_count = nullptr;
        double* d_carry = nullptr;
        double* d_wf_hit = nullptr;
        uint64_t* d_vis_pairs = nullptr;
        uint32_t* d_vis_count = nullptr;
        double* d_rays = nullptr;        // shadow-ray queue (wavefront pipeline only)
        uint32_t* d_ray_count = nullptr;  // [0] rays in the queue, [1] claim position of the persistent waves
        uint8_t* d_ray_flag = nullptr;
        uint64_t* d_ray_pairs = nullptr;
        uint32_t* d_wf_trace_q = nullptr;
        uint32_t* d_wf_count = nullptr;
        uint32_t* d_pool_next = nullptr;  // [0] eye, [1] light: pages of the deep vertex slots asked for in the current batch; [2] the batch cannot be completed
        double* d_fb_lane = nullptr;      // what the current batch adds to the sum buffers (fb_lane_slots planar buffers), see k_commit_batch
        unsigned long long* d_counters = nullptr;  // and to the counters
        uint32_t* h_result = nullptr;     // pinned ring of kResultRing x 4 words: d_pool_next as every batch left it
        uint64_t seq = 0;                 // batches enqueued on this lane
        std::vector<TimedLaunch> pending;  // timed launches of this lane not yet folded into the statistics
        // one marker event per enqueued batch with pending.size() at that point: waiting for the oldest marker
        // bounds the host's lead over the device without draining the other lanes
        struct Marker {   // event, pending.size(), pixel-samples of the batch as the progress counts them; the batch; where its result lands
            hipEvent_t first; size_t second; uint64_t paths;
            BatchDesc desc; const uint32_t* result;
        };
        std::deque<Marker> inflight;
    };
    static constexpr uint32_t kResultRing = 128;   // more than the batches a lane may have in flight (throttle: at most 64)
    // Batches that could not be completed (a subpath found the pool of deep vertex slots empty): their sums were dropped on
    // the device; they are rendered again in pieces small enough for the pool, and the batches after them are made smaller.
    struct Failed { BatchDesc desc; uint32_t asked_eye, asked_light; };
    std::vector<Failed> redo;
    uint64_t batch_limit = 0;      // pixel-samples per batch from now on (<= cap_paths; lowered when a batch failed)
    uint64_t safe_paths = 0;       // a batch of at most this many pixel-samples cannot exhaust the pool (pool pages / page levels)
    uint32_t page_levels = 0;
    uint32_t pool_pages = 0;
    uint32_t fb_lane_slots = 1;    // sum buffers per lane: the passes a batch may touch when they are kept apart, else one
    uint64_t batches_redone = 0;
    double tree_build_seconds = 0;   // buildLbvh alone (UBDPT_CREATE_TIMES prints the phases of ubdpt_create)
    std::vector<BatchLane> lanes;
    uint64_t next_batch = 0;      // batches enqueued so far; batch k runs on lane k % lanes.size()
    // ubdpt_query_progress reads these without the context mutex (an enqueue call holds that one while it waits for room on
    // the device): the counts are atomics, and the lanes' marker lists are changed and read under progress_mu only
    std::atomic<uint64_t> paths_enqueued{0};  // pixel-samples of all batches enqueued so far
    std::atomic<uint64_t> paths_held{0};      // pixel-samples of the passes held back for the next call
    std::mutex progress_mu;
    // Passes of an ubdpt_enqueue_passes call that did not fill a batch (the call's passes modulo the passes of a batch):
    // they are held back and start the next call's first batch when that call continues where this one ended; every
    // other call, and everything that waits for the enqueued work, launches them first.
    struct Held { uint64_t first = 0, stride = 1, seed = 0; uint32_t count = 0; } held;
    bool flushing_held = false;   // held passes are being launched: the call that took them promised them, a stop request does not drop them
    bool work_pending = false;    // batches were enqueued and not yet waited for
    double pending_since = 0;
    unsigned long long* d_counters = nullptr;

    std::atomic<bool> stop{false};
    std::mutex mu;
    ubdpt_stats stats{};
    unsigned long long kernel_counters[K_NUM][kNumCounters] = {{0}};
    double kernel_ms[K_NUM] = {0};
    double kernel_units[K_NUM] = {0};
    uint64_t kernel_launches[K_NUM] = {0};
    std::vector<TimedLaunch> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
    uint64_t stack_overflows = 0;
    uint64_t pool_overflows = 0;   // subpaths that found the pool of deep vertex slots empty

    hipError_t last_hip = hipSuccess;

    void beginTimed(int kernel, double units) { beginTimed(kernel, units, stream); }
    void endTimed() { endTimed(stream); }
    void beginTimed(int kernel, double units, hipStream_t stream) {
        TimedLaunch t;
        if (!event_pool.empty()) { t.start = event_pool.back().first; t.stop = event_pool.back().second; event_pool.pop_back(); }
        else { hipEventCreate(&t.start); hipEventCreate(&t.stop); }
        t.kernel = kernel; t.units = units;
        hipEventRecord(t.start, stream);
        pending.push_back(t);
    }
    void endTimed(hipStream_t stream) { hipEventRecord(pending.back().stop, stream); }
    // folds the first n entries of v (whose events have completed) into the per-kernel statistics
    void foldTimed(std::vector<TimedLaunch>& v, size_t n) {
        for (size_t k = 0; k < n && k < v.size(); k++) {
            TimedLaunch& t = v[k];
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
                kernel_ms[t.kernel] += ms; kernel_units[t.kernel] += t.units; kernel_launches[t.kernel]++;
            }
            event_pool.emplace_back(t.start, t.stop);
        }
        v.erase(v.begin(), v.begin() + std::min(n, v.size()));
    }
    void collectTimed() { foldTimed(pending, pending.size()); }  // call after the stream is synchronised
    void beginTimedLane(BatchLane& b, int kernel, double units) {
        TimedLaunch t;
        if (!event_pool.empty()) { t.start = event_pool.back().first; t.stop = event_pool.back().second; event_pool.pop_back(); }
        else { hipEventCreate(&t.start); hipEventCreate(&t.stop); }
        t.kernel = kernel; t.units = units;
        hipEventRecord(t.start, b.stream);
        b.pending.push_back(t);
    }
    void endTimedLane(BatchLane& b) { hipEventRecord(b.pending.back().stop, b.stream); }
    std::vector<hipEvent_t> marker_pool;
    void markBatch(BatchLane& b, uint64_t paths, const BatchDesc& desc, const uint32_t* result) {
        hipEvent_t ev;
        if (!marker_pool.empty()) { ev = marker_pool.back(); marker_pool.pop_back(); }
        else hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        hipEventRecord(ev, b.stream);
        std::lock_guard<std::mutex> pl(progress_mu);
        b.inflight.push_back(BatchLane::Marker{ev, b.pending.size(), paths, desc, result});
        paths_enqueued += paths;
    }
    // the lane's oldest batch has finished (its marker was reached): statistics, result, marker back to the pool
    void popFinished(BatchLane& b) {
        auto front = b.inflight.front();
        foldTimed(b.pending, front.second);
        if (front.result[2]) redo.push_back(Failed{front.desc, front.result[0], front.result[1]});
        std::lock_guard<std::mutex> pl(progress_mu);
        b.inflight.pop_front();
        for (auto& f : b.inflight) f.second -= front.second;
        marker_pool.push_back(front.first);
    }
    // waits until at most max_inflight batches of the lane are unfinished
    hipError_t throttle(BatchLane& b, size_t max_inflight) {
        while (b.inflight.size() > max_inflight) {
            hipError_t e = hipEventSynchronize(b.inflight.front().first);
            if (e != hipSuccess) return e;
            popFinished(b);
        }
        return hipSuccess;
    }
    // takes in the batches that have finished by now, without waiting
    void reapFinished() {
        for (auto& b : lanes)
            while (!b.inflight.empty() && hipEventQuery(b.inflight.front().first) == hipSuccess) popFinished(b);
        (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error status
    }
    // after every lane stream has been synchronised
    void collectLanes() {
        for (auto& b : lanes)
            while (!b.inflight.empty()) popFinished(b);
    }
    void freeAll() {
        hipSetDevice(device);
        if (stream) hipStreamSynchronize(stream);
        for (auto& t : pending) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
        for (auto& b : lanes) {
            if (b.stream) hipStreamSynchronize(b.stream);
            for (auto& t : b.pending) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
            for (auto& f : b.inflight) hipEventDestroy(f.first);
        }
        for (auto& e : marker_pool) hipEventDestroy(e);
        for (auto& f : fences) for (auto& e : f) hipEventDestroy(e);
        if (clear_ev) hipEventDestroy(clear_ev);
        for (auto& e : event_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        hipFree(d_objects); hipFree(d_emitters); hipFree(d_face_pos); hipFree(d_face_attr); hipFree(d_face_prob); hipFree(d_face_cdf);
        hipFree(d_textures); hipFree(d_texels); hipFree(d_nodes); hipFree(d_xnodes); hipFree(d_prims); hipFree(d_fb); hipFree(d_staging);
        if (h_staging) { hipHostFree(h_staging); h_staging = nullptr; }
        if (h_noalloc_counters) { hipHostFree(h_noalloc_counters); h_noalloc_counters = nullptr; }
        freeLanes();
        hipFree(d_counters);
        if (stream) hipStreamDestroy(stream);
    }
    void freeLanes() {
        for (auto& b : lanes) {
            if (b.stream) hipStreamSynchronize(b.stream);
            hipFree(b.E.v); hipFree(b.E.count); hipFree(b.E.page_tab); hipFree(b.L.v); hipFree(b.L.count); hipFree(b.L.page_tab); hipFree(b.d_pool_next);
            hipFree(b.d_fb_lane); hipFree(b.d_counters); if (b.h_result) hipHostFree(b.h_result);
            hipFree(b.d_ovf); hipFree(b.d_pairs); hipFree(b.d_pair_count); hipFree(b.d_carry); hipFree(b.d_wf_hit);
            hipFree(b.d_vis_pairs); hipFree(b.d_vis_count); hipFree(b.d_rays); hipFree(b.d_ray_count); hipFree(b.d_ray_flag); hipFree(b.d_ray_pairs); hipFree(b.d_wf_trace_q); hipFree(b.d_wf_count);
            if (b.stream) hipStreamDestroy(b.stream);
        }
        lanes.clear();
    }
};

namespace {

constexpr uint64_t kQueuePerPath = 20;            // entries of the pair / ray / visible-pair queues per pixel-sample of a batch
constexpr uint64_t kMaxGridLanes = 2048ull * 256;  // lanes of the largest grid a grid-stride kernel is launched with

#define HIPCK(ctx, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (ctx)->last_hip = e_; \
    fprintf(stderr, "ubdpt: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return UBDPT_HIP_ERROR; } } while (0)

// Vertex storage of one kind of subpath for batches of n pixel-samples: `dense` slots for every subpath, then the
// pool's pages, in one allocation; the page table; (silent: the caller retries with smaller batches when the memory does
// not suffice, and reports the failure that remains)
int allocStore(ubdpt_context* c, PathStore& P, uint64_t n, uint32_t maxv, uint32_t dense, uint32_t pool_pages, uint32_t* pool_next) {
    P.n = n; P.maxv = maxv; P.count_stride = 1;
    P.dense = dense; P.pool_pages = pool_pages; P.cap = n; P.pool_next = pool_next;
    P.pool_off = (uint64_t)kVertexDoubles * dense * n;
    const uint32_t levels = dense < maxv ? (maxv - dense + kPageSlots - 1) / kPageSlots : 0;
    hipError_t e = hipMalloc((void**)&P.v, sizeof(double) * (P.pool_off + (uint64_t)kVertexDoubles * kPageSlots * pool_pages));
    if (e == hipSuccess) e = hipMalloc((void**)&P.count, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void**)&P.page_tab, sizeof(uint32_t) * std::max<uint64_t>(1, (uint64_t)levels * n));
    if (e != hipSuccess) { c->last_hip = e; return UBDPT_HIP_ERROR; }
    return UBDPT_SUCCESS;
}

int flattenAndUpload(ubdpt_context* c, const ubdpt_scene_desc* sd) {
    std::vector<int32_t> emitters;
    std::vector<float> face_pos(9 * sd->num_faces);
    std::vector<FaceAttr> face_attr(sd->num_faces);
    std::vector<double> face_prob(sd->num_faces, 0.0);
    std::vector<double> face_cdf(sd->num_faces, 0.0);
    std::vector<DTexture> textures(sd->num_textures);
    std::vector<Prim> prims;

    for (uint32_t t = 0; t < sd->num_textures; t++) {
        const ubdpt_texture& in = sd->textures[t];
        DTexture& tx = textures[t];
        tx.kind = in.kind; tx.width = in.width; tx.height = in.height; tx.pad = 0;
        tx.color = mk3(in.color[0], in.color[1], in.color[2]);
        tx.texel_offset = in.texel_offset;
        if (in.kind == UBDPT_TEX_IMAGE) {
            uint64_t n = (uint64_t)in.width * in.height;
            if (n == 0 || in.texel_offset + 4 * n > sd->num_texel_bytes || !sd->texels) return UBDPT_INVALID_SCENE;
        } else if (in.kind != UBDPT_TEX_COLOR)
            return UBDPT_INVALID_SCENE;
    }
    for (uint64_t f = 0; f < sd->num_faces; f++) {
        for (int v = 0; v < 3; v++) {
            const ubdpt_mesh_vertex& mv = sd->vertices[3 * f + v];
            for (int k = 0; k < 3; k++) {
                face_pos[9 * f + 3 * v + k] = mv.pos[k];
                face_attr[f].n[3 * v + k] = mv.normal[k];
                face_attr[f].t[3 * v + k] = mv.tangent[k];
            }
            face_attr[f].uv[2 * v] = mv.uv[0];
            face_attr[f].uv[2 * v + 1] = mv.uv[1];
        }
        {   // the geometric normal of uapp/mesh.cpp:150, from the float corners (see FaceAttr)
            const float* p = &face_pos[9 * f];
            const d3 p0 = mk3((double)p[0], (double)p[1], (double)p[2]);
            const d3 p1 = mk3((double)p[3], (double)p[4], (double)p[5]);
            const d3 p2 = mk3((double)p[6], (double)p[7], (double)p[8]);
            const d3 ng = normalize(cross(p1 - p0, p2 - p0));
            face_attr[f].Ng[0] = ng.x; face_attr[f].Ng[1] = ng.y; face_attr[f].Ng[2] = ng.z;
            face_attr[f].pad = 0;
        }
    }
    c->h_objects.resize(sd->num_objects);
    for (uint32_t i = 0; i < sd->num_objects; i++) {
        const ubdpt_object& in = sd->objects[i];
        DObject& o = c->h_objects[i];
        memset(&o, 0, sizeof(o));
        if (in.kind != UBDPT_MODEL_SPHERE && in.kind != UBDPT_MODEL_MESH) return UBDPT_INVALID_SCENE;
        if (in.material > UBDPT_MAT_DIELECTRIC) return UBDPT_INVALID_SCENE;
        if (in.texture >= sd->num_textures) return UBDPT_INVALID_SCENE;
        o.kind = in.kind; o.material = in.material; o.texture = in.texture; o.is_emitter = in.is_emitter ? 1 : 0;
        o.mat_a = in.mat_a; o.mat_b = in.mat_b;
        memcpy(&o.W, in.W, sizeof(double) * 16);
        o.invW = inverse(o.W);  // uapp/object.cpp:8
        o.bsphere = identity4();
        o.power = mk3(in.power[0], in.power[1], in.power[2]);
        o.first_face = in.first_face; o.num_faces = in.num_faces;
        o.emitter_index = -1;
        if (o.kind == UBDPT_MODEL_MESH) {
            if (o.num_faces == 0 || o.first_face + o.num_faces > sd->num_faces || o.num_faces >= 0xffffffffULL)
                return UBDPT_INVALID_SCENE;
            const ubdpt_mesh_vertex* V = sd->vertices;
            // uapp/mesh.cpp:88-106
            double total_area = 0;
            for (uint64_t f = 0; f < o.num_faces; f++) {
                uint64_t g = o.first_face + f;
                double a = triangleArea(vpos(V[3 * g]), vpos(V[3 * g + 1]), vpos(V[3 * g + 2]));
                face_prob[g] = a;
                total_area += a;
            }
            for (uint64_t f = 0; f < o.num_faces; f++) face_prob[o.first_face + f] = face_prob[o.first_face + f] / total_area;
            double run = 0;
            for (uint64_t f = 0; f < o.num_faces; f++) { face_cdf[o.first_face + f] = run; run += face_prob[o.first_face + f]; }
            o.bsphere = boundingSphere(V, o.first_face, o.num_faces);
            for (uint64_t f = 0; f < o.num_faces; f++) {
                Prim p;
                memcpy(p.p, &face_pos[9 * (o.first_face + f)], sizeof(float) * 9);
                p.object = i; p.face = (uint32_t)f; for (uint32_t& x : p.pad) x = 0;
                prims.push_back(p);
            }
        } else {
            Prim p;
            memset(&p, 0, sizeof(p));
            p.object = i; p.face = kSphereFace;
            prims.push_back(p);
        }
        if (o.is_emitter) {
            o.emitter_index = (int32_t)emitters.size();
            emitters.push_back((int32_t)i);
            if (o.kind == UBDPT_MODEL_SPHERE) {  // uapp/implicitsphere.cpp:25-30
                double R = length(mul_xyz(o.W, mk3(1, 0, 0), 0.0));
                o.area = 4.0 * M_PI * R * R;
            } else {  // uapp/mesh.cpp:191-204
                double A = 0;
                const ubdpt_mesh_vertex* V = sd->vertices;
                for (uint64_t f = 0; f < o.num_faces; f++) {
                    uint64_t g = o.first_face + f;
                    A += triangleArea(mul_xyz(o.W, vpos(V[3 * g]), 1.0), mul_xyz(o.W, vpos(V[3 * g + 1]), 1.0),
                                      mul_xyz(o.W, vpos(V[3 * g + 2]), 1.0));
                }
                o.area = A;
            }
        }
    }
    // uengine/uscene.cpp:4-19
    {
        double total = 0;
        for (int32_t e : emitters) { d3 P = c->h_objects[e].power; total += (P.x + P.y + P.z) / c->h_objects[e].area; }
        for (int32_t e : emitters) {
            d3 P = c->h_objects[e].power;
            c->h_objects[e].prob = (P.x + P.y + P.z) / (total * c->h_objects[e].area);
        }
    }

    HIPCK(c, upload(&c->d_objects, c->h_objects));
    HIPCK(c, upload(&c->d_emitters, emitters));
    HIPCK(c, upload(&c->d_face_pos, face_pos));
    HIPCK(c, upload(&c->d_face_attr, face_attr));
    HIPCK(c, upload(&c->d_face_prob, face_prob));
    HIPCK(c, upload(&c->d_face_cdf, face_cdf));
    HIPCK(c, upload(&c->d_textures, textures));
    {
        std::vector<uint8_t> tex(sd->texels, sd->texels + (sd->texels ? sd->num_texel_bytes : 0));
        HIPCK(c, upload(&c->d_texels, tex));
    }
    Prim* d_unsorted = nullptr;
    HIPCK(c, upload(&d_unsorted, prims));
    BvhBuildResult br;
    const double tb0 = nowSeconds();
    hipError_t be = buildLbvh(c->d_objects, d_unsorted, (uint32_t)prims.size(), c->stream, &br);
    if (be == hipSuccess) be = hipStreamSynchronize(c->stream);
    c->tree_build_seconds = nowSeconds() - tb0;
    hipFree(d_unsorted);
    HIPCK(c, be);
    c->d_nodes = br.nodes;
    c->d_xnodes = br.xnodes;
    c->d_prims = br.prims;

    DScene& S = c->S;
    S.objects = c->d_objects; S.emitters = c->d_emitters; S.face_pos = c->d_face_pos; S.face_attr = c->d_face_attr;
    S.face_prob = c->d_face_prob; S.face_cdf = c->d_face_cdf; S.textures = c->d_textures; S.texels = c->d_texels;
    S.nodes = c->d_nodes; S.xnodes = c->d_xnodes; S.prims = c->d_prims;
    S.num_objects = sd->num_objects; S.num_emitters = (uint32_t)emitters.size(); S.num_prims = (uint32_t)prims.size();
    S.accel = c->opt.accel; S.num_faces = sd->num_faces;
    S.node_stride = (uint32_t)sizeof(BvhNode); S.xnode_stride = (uint32_t)sizeof(ExactNode);

    // uengine/ucamera.cpp:3-14,67-70 and uengine/ubdptrenderer.cpp:7-33
    const ubdpt_camera& cam = sd->camera;
    S.V = lookAtLH(mk3(cam.pos[0], cam.pos[1], cam.pos[2]), mk3(cam.look_at[0], cam.look_at[1], cam.look_at[2]), mk3(0, 1, 0));
    S.invV = inverse(S.V);
    S.image_plane_ratio = cam.ratio;
    S.image_plane_distance = 1.0 / tanf(cam.vfov / 2.0);  // float tan, as the reference
    const ubdpt_render_params& rp = c->params;
    S.img_res_x = (uint32_t)rp.img_res_x; S.img_res_y = (uint32_t)rp.img_res_y;
    S.num_pixel_strata = (uint32_t)(rp.pixel_subdiv * rp.pixel_subdiv);
    S.num_lens_strata = (uint32_t)(rp.lens_subdiv * rp.lens_subdiv);
    S.focus_plane_distance = rp.focus_plane_distance;
    S.lens_radius = rp.lens_size;
    S.min_depth = (uint32_t)rp.min_depth;
    S.max_path_vertices = c->opt.max_path_vertices;
    S.image_plane_area = 4 * S.image_plane_ratio;
    S.pixel_width = 2.0 * S.image_plane_ratio / (double)S.img_res_x;
    S.pixel_height = 2.0 / (double)S.img_res_y;
    S.lens_area = 0.0;  // never assigned in the reference (ubdptrenderer.cpp:29-30) and value-initialised (uengine.cpp:51)
    return UBDPT_SUCCESS;
}

int fetchCounters(ubdpt_context* c) {
    unsigned long long hk[K_NUM][kNumCounters];
    HIPCK(c, hipMemcpyAsync(hk, c->d_counters, sizeof(hk), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemsetAsync(c->d_counters, 0, sizeof(hk), c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < K_NUM; k++) {
        const unsigned long long* h = hk[k];
        for (int j = 0; j < kNumCounters; j++) {
            if (j == C_MAX_STACK) c->kernel_counters[k][j] = std::max(c->kernel_counters[k][j], h[j]);
            else c->kernel_counters[k][j] += h[j];
        }
        if (k <= kLastRenderKernel) {  // the probe kernels do not count as rendering work
            c->stats.closest_hit_rays += h[C_CLOSEST]; c->stats.shadow_rays += h[C_SHADOW]; c->stats.eye_vertices += h[C_EYE_V];
            c->stats.light_vertices += h[C_LIGHT_V]; c->stats.connections += h[C_CONN]; c->stats.splats += h[C_SPLAT];
            c->stats.bvh_nodes_visited += h[C_NODES]; c->stats.prims_tested += h[C_PRIMS]; c->stats.capped_subpaths += h[C_CAPPED];
        }
        c->stack_overflows += h[C_STACK_OVERFLOW];
        c->pool_overflows += h[C_POOL_OVERFLOW];
    }
    return UBDPT_SUCCESS;
}

}  // namespace

extern "C" {

int ubdpt_version(void) { return 100; }

const char* ubdpt_status_string(int status) {
    switch (status) {
    case UBDPT_SUCCESS: return "success";
    case UBDPT_UNINITIALIZED: return "uninitialized";
    case UBDPT_INVALID_SCENE: return "invalid scene";
    case UBDPT_INVALID_FORMAT: return "invalid format";
    case UBDPT_STOPPED: return "stopped";
    case UBDPT_NO_DATA: return "no data";
    case UBDPT_ERROR: return "error";
    case UBDPT_INVALID_ARG: return "invalid argument";
    case UBDPT_NO_DEVICE: return "no HIP device";
    case UBDPT_HIP_ERROR: return "HIP runtime error";
    }
    return "unknown status";
}

int ubdpt_device_count(int* count) {
    if (!count) return UBDPT_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return UBDPT_NO_DEVICE; }
    *count = n;
    return UBDPT_SUCCESS;
}

int ubdpt_create(const ubdpt_scene_desc* scene, const ubdpt_render_params* params, const ubdpt_options* options,
                 int device, ubdpt_context** out) {
    if (!out) return UBDPT_INVALID_ARG;
    *out = nullptr;
    if (!scene || !params) return UBDPT_INVALID_ARG;
    if (params->img_res_x == 0 || params->img_res_y == 0 || params->pixel_subdiv == 0 || params->lens_subdiv == 0 ||
        params->img_res_x > 65536 || params->img_res_y > 65536)
        return UBDPT_INVALID_ARG;
    if ((scene->num_objects && !scene->objects) || (scene->num_faces && !scene->vertices) ||
        (scene->num_textures && !scene->textures))
        return UBDPT_INVALID_SCENE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return UBDPT_NO_DEVICE;
    if (device < 0 || device >= ndev) return UBDPT_INVALID_ARG;

    ubdpt_context* c = new (std::nothrow) ubdpt_context();
    if (!c) return UBDPT_ERROR;
    double t0 = nowSeconds();
    c->device = device;
    if (options) c->opt = *options;
    if (c->opt.pipeline > 1) { delete c; return UBDPT_INVALID_ARG; }
    if (c->opt.max_path_vertices == 0) c->opt.max_path_vertices = 32;
    if (c->opt.max_path_vertices < 2) c->opt.max_path_vertices = 2;
    if (c->opt.max_path_vertices > 255) c->opt.max_path_vertices = 255;
    c->params = *params;
    int st = UBDPT_SUCCESS;
    double t_scene = t0;
    do {
        if (hipSetDevice(device) != hipSuccess) { st = UBDPT_HIP_ERROR; break; }
        if (hipStreamCreate(&c->stream) != hipSuccess) { st = UBDPT_HIP_ERROR; break; }
        if ((st = flattenAndUpload(c, scene)) != UBDPT_SUCCESS) break;
        t_scene = nowSeconds();
        c->npix = params->img_res_x * params->img_res_y;
        // passes in flight: enough paths to fill 256 CUs several times over, bounded by vertex storage
        // (per pixel-sample: vertex records and count words of both subpaths, pair / ray / visible-pair queues of
        // kQueuePerPath entries, two carry and hit records, queue words; the per-path kernels of pipeline 1 also need an
        // overflow stack per pixel-sample, the wavefront kernels one per lane of their largest grid)
        const bool wavefront = c->opt.pipeline == 0;
        // vertex slots: the first `dense` of every subpath, the rest in pages from a pool (paths.h)
        const uint32_t maxv = c->opt.max_path_vertices;
        uint32_t dense = c->opt.dense_vertex_slots ? c->opt.dense_vertex_slots : 16;
        dense = (dense + kPageSlots - 1) / kPageSlots * kPageSlots;
        if (dense >= maxv) dense = maxv;
        const uint32_t page_levels = dense < maxv ? (maxv - dense + kPageSlots - 1) / kPageSlots : 0;
        const uint32_t pages_per_k = dense < maxv ? (c->opt.vertex_pool_pages ? c->opt.vertex_pool_pages : 128u) : 0u;
        c->dense_slots = dense; c->pages_per_k = pages_per_k; c->page_levels = page_levels;
        uint64_t per_path = (uint64_t)(sizeof(double) * kVertexDoubles) * dense * 2 +
                            (uint64_t)(sizeof(double) * kVertexDoubles) * kPageSlots * pages_per_k * 2 / 1024 + 8ull * page_levels + 8 +
                            kQueuePerPath * (8 + 8 + 8 + sizeof(double) * kRayDoubles + 1) +
                            2 * sizeof(double) * (kCarryDoubles + kHitDoubles) + 16 + (wavefront ? 0 : 4ull * kOverflowStack) +
                            3 * sizeof(double);   // (the lane's sum buffers when the passes are kept apart)
        const uint64_t lane_fixed = wavefront ? 4ull * kOverflowStack * kMaxGridLanes : 0;
        auto laneBytes = [&](uint32_t passes) { return (per_path + 64) * c->npix * passes + lane_fixed; };
        uint32_t pif = c->opt.passes_in_flight;
        if (pif == 0) {
            // measured: a scene whose tree sits in LDS runs the same from 4 M pixel-samples per batch up (512 x 512 box,
            // 16 to 32 passes); a scene in HBM gains with every pass the memory holds (2 M triangles at 1080p: 2 / 3 / 4
            // passes per batch 266 / 279 / 286 Mpaths/s), because every launch of its ray tracing kernels ends with the
            // drain of its longest rays
            // (with the deep vertex slots in a pool six 1080p passes fit a lane: 2 x 4 / 2 x 6 / 2 x 8 passes 303 / 313 / 314)
            uint64_t target = sceneLdsLevel(c->S) == 2 ? 1ull << 22 : 3ull << 22;
            pif = (uint32_t)((target + c->npix - 1) / c->npix);
            if (pif < 1) pif = 1;
            if (pif > 64) pif = 64;
        }
        uint32_t nstreams = c->opt.streams ? std::min<uint32_t>(c->opt.streams, 16) : 2;
        // the batch lanes together may take four fifths of the memory that is free now (288 GB parts: ~230 GB, which holds
        // batches of four 1080p passes per lane; with half of the memory, two passes per lane, the 2 M-triangle scene
        // runs at 266 instead of 286 Mpaths/s). A second context on the same GPU gets four fifths of what is left, and
        // so on: smaller batches, same results. UBDPT_MEM_FRACTION overrides the share.
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 48ull << 30;
        // UBDPT_ASSUME_FREE_BYTES: what to take for the free memory instead (tests of the path below that shrinks the
        // batches when the snapshot was too optimistic, as it is when two contexts are created at the same moment)
        if (const char* e = getenv("UBDPT_ASSUME_FREE_BYTES")) { const double v = atof(e); if (v >= 1.0) mem_free = (size_t)v; }
        static const double mem_fraction = [] { const char* e = getenv("UBDPT_MEM_FRACTION"); double f = e ? atof(e) : 0.8; return f > 0.05 && f <= 0.95 ? f : 0.8; }();
        const uint64_t lane_budget = std::max<uint64_t>(1ull << 30, (uint64_t)((double)mem_free * mem_fraction) / nstreams);
        while (pif > 1 && laneBytes(pif) > lane_budget) pif--;
        // the packed queue entries hold a subpath id in 29 bits
        while (pif > 1 && 2 * c->npix * pif >= (1ull << 29) - 1) pif--;
        // one pass of a very large image may already exceed a lane's share: then a single lane gets the whole budget
        if (laneBytes(pif) > lane_budget) nstreams = 1;
        c->passes_in_flight = pif;
        c->cap_paths = c->npix * pif;
        // a batch smaller than one pass: a range of pixels of one pass (never across a pass boundary)
        if (c->opt.batch_pixel_samples && c->opt.batch_pixel_samples < c->npix) {
            c->passes_in_flight = 1;
            c->cap_paths = std::max<uint64_t>(4096, (c->opt.batch_pixel_samples + 255) / 256 * 256);
        }
        if (2 * c->cap_paths >= (1ull << 29) - 1) { st = UBDPT_INVALID_ARG; break; }
        // bounces run as wavefront launches before the per-lane tail kernel takes the survivors: every launch of the
        // persistent trace kernel ends with the tail of its longest ray, which the last, thin bounces do not pay back
        // on scenes in HBM (2 M triangles: depth 6 / 7 / 8 = 260 / 255 / 253 Mpaths/s); a scene in LDS prefers 8
        c->wf_depth = c->opt.wavefront_depth ? std::min<uint32_t>(c->opt.wavefront_depth, kMaxWfDepth)
                                             : (sceneLdsLevel(c->S) == 2 ? 8u : 6u);
        c->pass_slots = std::min<uint32_t>(c->opt.pass_slots, 64);
        c->ovf_cap = c->opt.traversal_stack_entries ? std::min<uint32_t>(c->opt.traversal_stack_entries, kLdsStack + kOverflowStack) : kLdsStack + kOverflowStack;
        for (int& f : c->slot_fence) f = -1;
        auto ck = [&](hipError_t e) { if (e != hipSuccess && st == UBDPT_SUCCESS) { c->last_hip = e; st = UBDPT_HIP_ERROR; } };
        ck(hipMalloc((void**)&c->d_fb, sizeof(double) * 3 * c->npix * (1 + c->pass_slots)));
        ck(hipMalloc((void**)&c->d_staging, sizeof(double) * 3 * c->npix));
        if (st != UBDPT_SUCCESS) break;
        ck(hipMemsetAsync(c->d_fb, 0, sizeof(double) * 3 * c->npix * (1 + c->pass_slots), c->stream));
        // The lanes' buffers. hipMemGetInfo above is a snapshot: when another process took memory in between, the
        // allocation is tried again with batches of half the passes, then with one lane.
        for (int attempt = 0;; attempt++) {
        c->ovf_threads = ((c->cap_paths + 255) / 256) * 256;
        c->ovf_lanes = wavefront ? std::min<uint64_t>(c->ovf_threads, kMaxGridLanes) : c->ovf_threads;
        // pair / ray / visible-pair queues: kQueuePerPath entries per pixel-sample (a box scene averages 6 to 10); pixel-samples
        // beyond the queue's end are finished by k_pair_overflow. UBDPT_PAIR_QUEUE_PER_PATH shrinks it for tests of that path.
        {
            const char* e = getenv("UBDPT_PAIR_QUEUE_PER_PATH");
            const unsigned long long per_path_entries = e && atoi(e) > 0 ? (unsigned long long)atoi(e) : kQueuePerPath;
            c->pair_capacity = c->cap_paths * std::min<unsigned long long>(per_path_entries, kQueuePerPath);
        }
        c->lanes.resize(nstreams);
        for (auto& b : c->lanes) {
            ck(hipStreamCreate(&b.stream));
            if (st != UBDPT_SUCCESS) break;
            ck(hipMalloc((void**)&b.d_pool_next, 4 * sizeof(uint32_t)));
            if (st != UBDPT_SUCCESS) break;
            ck(hipMemsetAsync(b.d_pool_next, 0, 4 * sizeof(uint32_t), c->stream));
            // the pool: vertex_pool_pages per 1024 subpaths of a batch, and never less than 1024 subpaths can fill whatever
            // happens (a batch that exhausts the pool is rendered again in pieces, which must end somewhere).
            // UBDPT_POOL_PAGES sets the number itself, for tests of that path.
            uint32_t pool_pages = c->pages_per_k ? (uint32_t)std::max<uint64_t>(1024ull * c->page_levels, c->cap_paths * c->pages_per_k / 1024) : 0u;
            if (const char* e = getenv("UBDPT_POOL_PAGES")) { if (c->pages_per_k && atoi(e) > 0) pool_pages = std::max<uint32_t>((uint32_t)atoi(e), c->page_levels); }
            c->pool_pages = pool_pages;
            c->safe_paths = c->page_levels ? pool_pages / c->page_levels : c->cap_paths;
            // what a batch adds to the sum buffers and to the counters is collected per lane and handed on when the batch
            // is complete: one buffer, or one per pass of a batch when the context keeps passes apart
            c->fb_lane_slots = c->pass_slots ? std::max<uint32_t>(1, c->passes_in_flight) : 1;
            ck(hipMalloc((void**)&b.d_fb_lane, sizeof(double) * 3 * c->npix * c->fb_lane_slots));
            ck(hipMalloc((void**)&b.d_counters, sizeof(unsigned long long) * kNumCounters * K_NUM));
            ck(hipHostMalloc((void**)&b.h_result, sizeof(uint32_t) * 4 * ubdpt_context::kResultRing, hipHostMallocDefault));
            if (st != UBDPT_SUCCESS) break;
            ck(hipMemsetAsync(b.d_fb_lane, 0, sizeof(double) * 3 * c->npix * c->fb_lane_slots, c->stream));
            ck(hipMemsetAsync(b.d_counters, 0, sizeof(unsigned long long) * kNumCounters * K_NUM, c->stream));
            memset(b.h_result, 0, sizeof(uint32_t) * 4 * ubdpt_context::kResultRing);
            if ((st = allocStore(c, b.E, c->cap_paths, c->opt.max_path_vertices, c->dense_slots, pool_pages, b.d_pool_next)) != UBDPT_SUCCESS) break;
            if ((st = allocStore(c, b.L, c->cap_paths, c->opt.max_path_vertices, c->dense_slots, pool_pages, b.d_pool_next + 1)) != UBDPT_SUCCESS) break;
            uint64_t n2 = 2 * c->cap_paths;
            ck(hipMalloc((void**)&b.d_ovf, sizeof(int32_t) * kOverflowStack * c->ovf_lanes));
            ck(hipMalloc((void**)&b.d_pairs, sizeof(uint64_t) * c->pair_capacity));
            ck(hipMalloc((void**)&b.d_pair_count, sizeof(unsigned long long)));
            ck(hipMalloc((void**)&b.d_carry, sizeof(double) * kCarryDoubles * n2));
            ck(hipMalloc((void**)&b.d_wf_hit, sizeof(double) * kHitDoubles * n2));
            ck(hipMalloc((void**)&b.d_vis_pairs, sizeof(uint64_t) * c->pair_capacity));
            ck(hipMalloc((void**)&b.d_vis_count, sizeof(uint32_t)));
            ck(hipMalloc((void**)&b.d_ray_count, 4 * sizeof(uint32_t)));
            if (c->opt.pipeline == 0) {
                ck(hipMalloc((void**)&b.d_rays, sizeof(double) * kRayDoubles * c->pair_capacity));
                ck(hipMalloc((void**)&b.d_ray_flag, c->pair_capacity));
                ck(hipMalloc((void**)&b.d_ray_pairs, sizeof(uint64_t) * c->pair_capacity));
            }
            ck(hipMalloc((void**)&b.d_wf_trace_q, sizeof(uint32_t) * 2 * n2));
            ck(hipMalloc((void**)&b.d_wf_count, sizeof(uint32_t) * kWfQStride * (kMaxWfDepth + 2)));
            if (st != UBDPT_SUCCESS) break;
        }
        if (st == UBDPT_SUCCESS) break;
        // out of memory: batches of half the passes, then one lane, then batches of a part of a pass (down to 64 K
        // pixel-samples); the results do not depend on the batch size
        const bool can_shrink = c->passes_in_flight > 1 || nstreams > 1 || c->cap_paths > (1ull << 16);
        if (c->last_hip != hipErrorOutOfMemory || !can_shrink || attempt >= 24) {
            fprintf(stderr, "ubdpt: HIP error %s while allocating the batch buffers (%llu pixel-samples per batch, %u lane(s))\n",
                    hipGetErrorString(c->last_hip), (unsigned long long)c->cap_paths, nstreams);
            break;
        }
        c->freeLanes();
        (void)hipGetLastError();
        st = UBDPT_SUCCESS;
        if (c->passes_in_flight > 1 && c->cap_paths >= c->npix) {
            c->passes_in_flight = (c->passes_in_flight + 1) / 2;
            c->cap_paths = c->npix * c->passes_in_flight;
        } else if (nstreams > 1) {
            nstreams = 1;
        } else {
            c->passes_in_flight = 1;
            c->cap_paths = std::max<uint64_t>(1ull << 16, ((c->cap_paths / 2) + 255) / 256 * 256);
        }
        }
        if (st != UBDPT_SUCCESS) break;
        c->batch_limit = c->cap_paths;
        // set-up of the entry point that allocates nothing: this GPU's address of the static scratch array (asked for with
        // this GPU current) and a pinned word block its counters are copied to
        ck(noAllocScratch(device, &c->noalloc_scratch));
        ck(hipHostMalloc((void**)&c->h_noalloc_counters, sizeof(unsigned long long) * kNumCounters, hipHostMallocDefault));
        ck(hipMalloc((void**)&c->d_counters, sizeof(unsigned long long) * kNumCounters * K_NUM));
        if (st != UBDPT_SUCCESS) break;
        ck(hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long) * kNumCounters * K_NUM, c->stream));
        ck(hipStreamSynchronize(c->stream));
    } while (0);
    if (st != UBDPT_SUCCESS) {
        c->freeAll();
        delete c;
        return st;
    }
    c->stats.build_seconds = nowSeconds() - t0;
    if (getenv("UBDPT_CREATE_TIMES"))
        fprintf(stderr, "ubdpt_create: %.3f s (scene flattened, uploaded and tree built %.3f s, of which the tree %.3f s; batch buffers %.3f s)\n",
                c->stats.build_seconds, t_scene - t0, c->tree_build_seconds, nowSeconds() - t_scene);
    *out = c;
    return UBDPT_SUCCESS;
}

int ubdpt_destroy(ubdpt_context* c) {
    if (!c) return UBDPT_INVALID_ARG;
    c->freeAll();
    delete c;
    return UBDPT_SUCCESS;
}

// A ray whose traversal stack was full dropped a subtree: whatever was computed with it may miss a hit.
// The call that finds the counter set fails; the counter is cleared so that the context stays usable.
static int reportStackOverflows(ubdpt_context* c) {
    if (!c->stack_overflows) return UBDPT_SUCCESS;
    fprintf(stderr, "ubdpt: traversal stack overflow in %llu waves or rays; the result of this call is not valid\n",
            (unsigned long long)c->stack_overflows);
    c->stack_overflows = 0;
    return UBDPT_ERROR;
}

// Waits for every enqueued batch, then folds the event timings and the device counters into the
// statistics. Called with the context mutex held.
static int renderPassesLocked(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                              uint32_t* passes_done, bool wait, bool separate, uint32_t first_slot, bool may_hold);
// launches the passes a call held back (context mutex held)
static int flushHeld(ubdpt_context* c) {
    if (!c->held.count) return UBDPT_SUCCESS;
    const ubdpt_context::Held h = c->held;
    c->held.count = 0; c->paths_held = 0;
    c->flushing_held = true;
    const int st = renderPassesLocked(c, h.first, h.count, h.stride, h.seed, nullptr, false, false, 0, false);
    c->flushing_held = false;
    return st;
}


// One batch: pixel-samples [d.pos, d.pos + d.n) of the call d.call, on the next lane.
static int launchBatch(ubdpt_context* c, const ubdpt_context::BatchDesc& d, bool redo) {
    const bool stats = c->opt.collect_stats != 0;
    const bool separate = d.call.separate;
    const uint64_t nbatch = d.n, pos = d.pos;
    uint64_t& batch = c->next_batch;
    ubdpt_context::BatchLane& B = c->lanes[batch % c->lanes.size()];
    hipStream_t sm = B.stream;
    RenderArgs A;
    A.S = c->S;
    A.E = B.E; A.L = B.L;
    A.E.n = A.L.n = nbatch;
    // the wavefront pipeline keeps the live count words inside the carry records (below); the stores' own dense
    // arrays take the count words of the finished subpaths
    A.fin_E = B.E.count; A.fin_L = B.L.count;
    // what the batch adds goes to the lane's own sum buffers (one, or one per pass the batch touches when the call keeps
    // its passes apart) and from there to the context's when the batch is complete
    const uint64_t pass0 = pos / c->npix;
    const uint32_t passes_touched = (uint32_t)((pos + nbatch - 1) / c->npix - pass0) + 1;
    if (separate && passes_touched > c->fb_lane_slots) return UBDPT_ERROR;   // (batches are whole passes or parts of one)
    A.fb = B.d_fb_lane;
    A.fb_slots = separate ? passes_touched : 0;
    A.first_pass = d.call.first_pass;
    A.path_offset = pos;
    A.batch_pass0 = pass0;
    A.batch_pixel0 = pos % c->npix;
    A.pass_stride = d.call.pass_stride;
    A.seed = d.call.seed;
    A.ovf = B.d_ovf; A.ovf_stride = A.E.n; A.ovf_cap = c->ovf_cap; A.pad_ovf = 0;
    A.counters = B.d_counters;
    A.batch_fail = B.d_pool_next + 2;
    A.pairs = B.d_pairs; A.pair_capacity = c->pair_capacity; A.pair_count = B.d_pair_count;
    A.carry = B.d_carry; A.wf_trace_q = B.d_wf_trace_q; A.wf_count = B.d_wf_count;
    A.wf_hit = B.d_wf_hit; A.vis_pairs = B.d_vis_pairs; A.vis_count = B.d_vis_count;
    A.rays = B.d_rays; A.ray_count = B.d_ray_count; A.ray_head = B.d_ray_count + 1; A.ray_flag = B.d_ray_flag; A.ray_pairs = B.d_ray_pairs;
    double units = (double)A.E.n;
    // grid-stride kernels: at most 2048 workgroups (8 per CU), never more lanes than the stack overflow buffer covers;
    // 512 .. 16384 workgroups measured within 2 % of each other
    unsigned pblocks = (unsigned)std::min<uint64_t>(kMaxGridLanes / 256, c->ovf_threads / 256);
    // the two persistent ray tracing kernels: what the GPU holds at once (256 CUs x 4 workgroups of 4 waves). A
    // larger grid only queues workgroups that find the ray queue empty; measured on the 2 M-triangle scene:
    // 2048 workgroups 234, 1024: 239, 768: 231, 512: 225 Mpaths/s. UBDPT_TRACE_BLOCKS overrides it for measurements.
    static const unsigned tblocks_env = [] { const char* e = getenv("UBDPT_TRACE_BLOCKS"); return e ? (unsigned)atoi(e) : 0u; }();
    const unsigned tblocks = std::min(tblocks_env ? tblocks_env : 1024u, pblocks);
    const uint64_t pstride = c->ovf_lanes;  // the wavefront kernels' grids have at most pblocks * 256 lanes
    // pipeline: 0 = wavefront, small kernels: gen, D x (trace, shade), tail; pair enum, setup, rays, compact, eval (default)
    //           1 = per-path subpath kernels + per-path connection loop (the reference's loop structure, kept as a cross-check)
    const uint32_t pl = c->opt.pipeline;
    if (pl == 0) {
        // the count word of subpath `id` is the eighth double of its carry record's first line (eye subpaths
        // first, then the light subpaths), so trace and shade touch one line less per subpath
        const uint32_t stride = kCarryDoubles * (uint32_t)(sizeof(double) / sizeof(uint32_t));
        A.E.count = reinterpret_cast<uint32_t*>(B.d_carry + kCarryCountSlot);
        A.L.count = reinterpret_cast<uint32_t*>(B.d_carry + A.E.n * kCarryDoubles + kCarryCountSlot);
        A.E.count_stride = A.L.count_stride = stride;
    }
    const uint32_t D = c->wf_depth;
    auto run = [&](int kernel, double u, auto&& launch) {
        A.counters = B.d_counters + kernel * kNumCounters;
        c->beginTimedLane(B, kernel, u); launch(); c->endTimedLane(B);
    };
    HIPCK(c, hipMemsetAsync(B.d_pool_next, 0, 4 * sizeof(uint32_t), sm));
    if (pl == 0) {
        HIPCK(c, hipMemsetAsync(B.d_wf_count, 0, sizeof(uint32_t) * kWfQStride * (D + 2), sm));
        A.ovf_stride = pstride;
        run(K_WF_GEN, 2 * units, [&] { launchWfGen(A, stats, sm); });
        for (uint32_t dd = 0; dd < D; dd++) {
            run(K_WF_TRACE, dd == 0 ? units : 0, [&] { launchWfTrace(A, dd, tblocks, pblocks, stats, sm); });
            run(K_WF_SHADE, dd == 0 ? units : 0, [&] { launchWfShade(A, dd, pblocks, stats, sm); });
        }
        run(K_WF_TAIL, units, [&] { launchWfTail(A, D, pblocks, stats, sm); });
        HIPCK(c, hipMemsetAsync(B.d_pair_count, 0, sizeof(unsigned long long), sm));
        HIPCK(c, hipMemsetAsync(B.d_vis_count, 0, sizeof(uint32_t), sm));
        HIPCK(c, hipMemsetAsync(B.d_ray_count, 0, 4 * sizeof(uint32_t), sm));
        run(K_PAIR_ENUM, units, [&] { launchPairEnum(A, pblocks, stats, sm); });
        if (sceneLdsLevel(A.S) == 2) {
            run(K_RAYS_VISIBLE, units, [&] { launchPairVisibleLds(A, pblocks, stats, sm); });
        } else {
            run(K_PAIR_SETUP, units, [&] { launchPairSetup(A, pblocks, stats, sm); });
            run(K_RAYS_VISIBLE, units, [&] { launchRaysVisible(A, tblocks, stats, sm); });
            run(K_PAIR_COMPACT, units, [&] { launchPairCompact(A, pblocks, sm); });
        }
        run(K_PAIR_EVAL, units, [&] { launchPairEval(A, pblocks, stats, sm); });
    } else {
        A.ovf_stride = A.E.n;
        run(K_EYE, units, [&] { launchEye(A, stats, sm); });
        run(K_LIGHT, units, [&] { launchLight(A, stats, sm); });
        run(K_CONNECT, units, [&] { launchConnect(A, stats, sm); });
    }
    // the batch's sums and counters move on to the context's (or are dropped, when the batch could not be completed), and
    // the host is told which of the two it was
    double* target = c->d_fb + (separate ? 3 * c->npix * (1 + (uint64_t)d.call.first_slot + pass0) : 0);
    launchCommitBatch(B.d_fb_lane, c->npix, separate ? passes_touched : 1, target, B.d_pool_next + 2, B.d_counters, c->d_counters,
                      (uint32_t)(kNumCounters * K_NUM), sm);
    uint32_t* result = B.h_result + 4 * (B.seq++ % ubdpt_context::kResultRing);
    HIPCK(c, hipMemcpyAsync(result, B.d_pool_next, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, sm));
    HIPCK(c, hipGetLastError());
    batch++;
    // keep at most a few batches (about 16 M pixel-samples) queued per lane, so that a stop request is
    // honoured promptly and the host does not run arbitrarily far ahead of the device
    // (the pieces of a batch that is rendered again were counted as progress when the batch was enqueued)
    c->markBatch(B, redo ? 0 : nbatch, d, result);
    const size_t max_inflight = (size_t)std::max<uint64_t>(2, std::min<uint64_t>(64, (16ull << 20) / std::max<uint64_t>(1, c->cap_paths)));
    HIPCK(c, c->throttle(B, max_inflight));
    return UBDPT_SUCCESS;
}

// Batches for the pixel-samples [pos, end) of a call, `pos` moving along. From pixel-sample `stop_from` on a stop request
// ends the range at the next pass boundary (UBDPT_STOPPED). Batches that failed meanwhile are rendered again on the way.
static int processRedo(ubdpt_context* c);
static int enqueueRange(ubdpt_context* c, const ubdpt_context::CallDesc& call, uint64_t& pos, uint64_t end, uint64_t stop_from,
                        bool redo = false) {
    while (pos < end) {
        if (!redo && pos % c->npix == 0 && pos >= stop_from && c->stop.load()) return UBDPT_STOPPED;
        // whole passes, or a range of pixels inside one pass (never across a pass boundary)
        uint64_t lim = std::min(c->cap_paths, c->batch_limit);
        if (lim >= c->npix) lim = lim / c->npix * c->npix;
        uint64_t nbatch = std::min<uint64_t>(lim, end - pos);
        if (lim < c->npix) nbatch = std::min<uint64_t>(nbatch, c->npix - pos % c->npix);
        ubdpt_context::BatchDesc d;
        d.call = call; d.pos = pos; d.n = nbatch;
        int st = launchBatch(c, d, redo);
        if (st != UBDPT_SUCCESS) return st;
        pos += nbatch;
        if (!redo && !c->redo.empty()) {
            st = processRedo(c);
            if (st != UBDPT_SUCCESS) return st;
        }
    }
    return UBDPT_SUCCESS;
}

// Renders the batches again that could not be completed, in pieces the pool of deep vertex slots can serve: the pages
// a failed batch asked for (a lower bound of what it needs: a subpath that was turned away asks for nothing more) say
// into how many pieces to cut it; a piece of safe_paths pixel-samples cannot fail. The batches that follow are made as
// small as the pieces: the scene is what exhausts the pool, and it stays.
static int processRedo(ubdpt_context* c) {
    while (!c->redo.empty()) {
        const ubdpt_context::Failed f = c->redo.back();
        c->redo.pop_back();
        if (f.desc.n <= c->safe_paths || c->pool_pages == 0) {
            fprintf(stderr, "ubdpt: a batch of %llu pixel-samples exhausted a pool of %u pages (internal error)\n",
                    (unsigned long long)f.desc.n, c->pool_pages);
            return UBDPT_ERROR;
        }
        const uint64_t asked = std::max(f.asked_eye, f.asked_light);
        const uint64_t pieces = std::max<uint64_t>(2, (3 * asked / 2 + c->pool_pages - 1) / c->pool_pages);
        uint64_t lim = std::max<uint64_t>(c->safe_paths, f.desc.n / pieces);
        if (lim >= c->npix) lim = lim / c->npix * c->npix;
        c->batch_limit = std::min(c->batch_limit, lim);
        c->batches_redone++;
        uint64_t pos = f.desc.pos;
        int st = enqueueRange(c, f.desc.call, pos, f.desc.pos + f.desc.n, UINT64_MAX, true);
        if (st != UBDPT_SUCCESS) return st;
    }
    return UBDPT_SUCCESS;
}

// Waits for everything that was enqueued, batches that have to be rendered again included.
static int settle(ubdpt_context* c) {
    for (;;) {
        for (auto& b : c->lanes) HIPCK(c, hipStreamSynchronize(b.stream));
        c->collectLanes();
        if (c->redo.empty()) return UBDPT_SUCCESS;
        int st = processRedo(c);
        if (st != UBDPT_SUCCESS) return st;
    }
}

static int finishPending(ubdpt_context* c) {
    if (!c->work_pending) return UBDPT_SUCCESS;
    { int fh = flushHeld(c); if (fh != UBDPT_SUCCESS) return fh; }
    { int ss = settle(c); if (ss != UBDPT_SUCCESS) return ss; }
    c->work_pending = false;
    c->collectTimed();
    int fc = fetchCounters(c);
    if (fc != UBDPT_SUCCESS) return fc;
    c->stats.render_seconds += nowSeconds() - c->pending_since;
    return reportStackOverflows(c);
}

static int renderPasses(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                        uint32_t* passes_done, bool wait, bool separate = false, uint32_t first_slot = 0) {
    if (passes_done) *passes_done = 0;
    if (!c) return UBDPT_INVALID_ARG;
    if (separate && (c->pass_slots == 0 || (uint64_t)first_slot + n_passes > c->pass_slots)) return UBDPT_INVALID_ARG;
    if (pass_stride == 0) pass_stride = 1;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    c->stop = false;  // UBDPTRenderer::renderPass resets m_stop (ubdptrenderer.cpp:37)
    return renderPassesLocked(c, first_pass, n_passes, pass_stride, seed, passes_done, wait, separate, first_slot, true);
}

static int renderPassesLocked(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                              uint32_t* passes_done, bool wait, bool separate, uint32_t first_slot, bool may_hold) {
    // passes an earlier call held back: this call takes them along when it continues where that call ended
    uint32_t merged = 0;
    if (c->held.count) {
        const ubdpt_context::Held h = c->held;
        if (may_hold && !separate && h.stride == pass_stride && h.seed == seed &&
            first_pass == h.first + (uint64_t)h.count * h.stride && n_passes <= 0xffffffffu - h.count) {
            c->held.count = 0; c->paths_held = 0;
            first_pass = h.first;
            n_passes += h.count;
            merged = h.count;
        } else {
            int fh = flushHeld(c);
            if (fh != UBDPT_SUCCESS) return fh;
        }
    }
    if (separate) {   // the per-pass buffers of this call start from zero
        if (wait) {
            int fs = finishPending(c);
            if (fs != UBDPT_SUCCESS) return fs;
        }
        // cleared on the copy stream, which is also the one that reads them back, so the clear follows every earlier
        // read of these buffers; the lanes wait for the clear, not for each other's older batches. Batches of an earlier
        // call that may still be adding into these buffers (nobody read them) are waited for first.
        bool seen[ubdpt_context::kFences] = {false};
        for (uint32_t s = first_slot; s < first_slot + n_passes; s++) {
            int f = c->slot_fence[s];
            if (f < 0 || seen[f]) continue;
            seen[f] = true;
            // (waited for on the host as well: one of those batches may have to be rendered again, see processRedo, and
            // that must not happen after the clear. When the buffers were read, their batches finished long ago)
            for (hipEvent_t ev : c->fences[f]) { HIPCK(c, hipEventSynchronize(ev)); HIPCK(c, hipStreamWaitEvent(c->stream, ev, 0)); }
        }
        c->reapFinished();
        if (!c->redo.empty()) { int ss = settle(c); if (ss != UBDPT_SUCCESS) return ss; }
        HIPCK(c, hipMemsetAsync(c->d_fb + 3 * c->npix * (1 + (uint64_t)first_slot), 0, sizeof(double) * 3 * c->npix * n_passes, c->stream));
        if (!c->clear_ev) HIPCK(c, hipEventCreateWithFlags(&c->clear_ev, hipEventDisableTiming));
        HIPCK(c, hipEventRecord(c->clear_ev, c->stream));
        for (auto& b : c->lanes) HIPCK(c, hipStreamWaitEvent(b.stream, c->clear_ev, 0));
    }
    if (!c->work_pending) { c->work_pending = true; c->pending_since = nowSeconds(); }
    uint32_t done = 0;
    int status = UBDPT_SUCCESS;
    // pixel-samples of this call in (pass, pixel) order; a batch is a contiguous range of them: whole passes, or
    // (when the batch size is below one pass) a range of pixels inside one pass
    // a call that is not waited for leaves the passes that do not fill a batch to the next call (see Held), provided
    // it launches at least one batch itself, so that the device has work until then
    uint32_t hold = 0;
    if (may_hold && !wait && !separate && c->cap_paths >= c->npix && c->passes_in_flight > 1 && n_passes > c->passes_in_flight)
        hold = n_passes % c->passes_in_flight;
    const uint64_t total_paths = (uint64_t)(n_passes - hold) * c->npix;
    ubdpt_context::CallDesc call;
    call.first_pass = first_pass; call.pass_stride = pass_stride; call.seed = seed; call.separate = separate; call.first_slot = first_slot;
    uint64_t pos = 0;
    // (the passes taken over from an earlier call were promised by that call: a stop request does not drop them)
    const uint64_t stop_from = c->flushing_held ? UINT64_MAX : (uint64_t)merged * c->npix;
    {
        int es = enqueueRange(c, call, pos, total_paths, stop_from);
        if (es == UBDPT_STOPPED) status = es;
        else if (es != UBDPT_SUCCESS) return es;
    }
    done = (uint32_t)(pos / c->npix);
    c->stats.passes += done;
    c->stats.pixel_samples += (uint64_t)done * c->npix;
    if (hold && status == UBDPT_SUCCESS) {
        c->held.first = first_pass + (uint64_t)(n_passes - hold) * pass_stride;
        c->held.count = hold; c->held.stride = pass_stride; c->held.seed = seed;
        c->paths_held = (uint64_t)hold * c->npix;
    } else
        hold = 0;
    // passes of THIS call that were launched or are held for the next one
    if (passes_done) *passes_done = done + hold - merged;
    if (separate) {
        int fence = -1;
        if (!wait) {  // one event per lane behind the batches of this call
            fence = (int)(c->next_fence++ % ubdpt_context::kFences);
            auto& ev = c->fences[fence];
            while (ev.size() < c->lanes.size()) {
                hipEvent_t e;
                HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ev.push_back(e);
            }
            for (size_t l = 0; l < c->lanes.size(); l++) HIPCK(c, hipEventRecord(ev[l], c->lanes[l].stream));
        }
        for (uint32_t s = first_slot; s < first_slot + n_passes; s++) c->slot_fence[s] = fence;
    }
    if (wait) {
        int fs = finishPending(c);
        if (fs != UBDPT_SUCCESS) return fs;
    }
    return status;
}

int ubdpt_render_passes(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                        uint32_t* passes_done) {
    return renderPasses(c, first_pass, n_passes, pass_stride, seed, passes_done, true);
}

int ubdpt_enqueue_passes(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                         uint32_t* passes_enqueued) {
    return renderPasses(c, first_pass, n_passes, pass_stride, seed, passes_enqueued, false);
}

int ubdpt_query_progress(ubdpt_context* c, uint64_t* samples_enqueued, uint64_t* samples_finished) {
    if (!c) return UBDPT_INVALID_ARG;
    // not the context mutex: an enqueue call holds it while it waits for room on the device, and a thread that polls
    // the progress must not wait with it
    std::lock_guard<std::mutex> lock(c->progress_mu);
    if (hipSetDevice(c->device) != hipSuccess) return UBDPT_HIP_ERROR;
    uint64_t unfinished = 0;
    for (auto& b : c->lanes) {
        // a lane's batches finish in order: everything from the first marker that has not been reached is unfinished
        bool reached = true;
        for (auto& m : b.inflight) {
            if (reached && hipEventQuery(m.first) != hipSuccess) reached = false;
            if (!reached) unfinished += m.paths;
        }
    }
    (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error status
    // passes held back for the next call count as enqueued: the call that took them reported them so
    const uint64_t enq = c->paths_enqueued.load();
    if (samples_enqueued) *samples_enqueued = enq + c->paths_held.load();
    if (samples_finished) *samples_finished = enq - unfinished;
    return UBDPT_SUCCESS;
}

int ubdpt_render_passes_separate(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride,
                                 uint64_t seed, uint32_t* passes_done) {
    return renderPasses(c, first_pass, n_passes, pass_stride, seed, passes_done, true, true);
}

int ubdpt_enqueue_passes_separate(ubdpt_context* c, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride,
                                  uint64_t seed, uint32_t first_slot, uint32_t* passes_enqueued) {
    return renderPasses(c, first_pass, n_passes, pass_stride, seed, passes_enqueued, false, true, first_slot);
}

int ubdpt_synchronize(ubdpt_context* c) {
    if (!c) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    return finishPending(c);
}

int ubdpt_request_stop(ubdpt_context* c) {
    if (!c) return UBDPT_INVALID_ARG;
    c->stop = true;
    return UBDPT_SUCCESS;
}

int ubdpt_read_sum_device(ubdpt_context* c, double* device_rgb) {
    if (!c || !device_rgb) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    launchInterleave(c->d_fb, c->npix, device_rgb, c->stream);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    return UBDPT_SUCCESS;
}

int ubdpt_accumulate_sum_device(ubdpt_context* c, double* device_rgb) {
    if (!c || !device_rgb) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    launchInterleaveAdd(c->d_fb, c->npix, device_rgb, c->stream);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    return UBDPT_SUCCESS;
}

// buffer 0 is the sum buffer, buffer 1 + k the per-pass buffer k
// out[i] = base[i] + add[i] over n doubles on up to `threads` host threads (contiguous ranges; one thread below a
// million doubles, where starting threads costs more than the loop). base may be out.
static void addOnto(const double* base, const double* add, double* out, uint64_t n, unsigned threads) {
    auto body = [=](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) out[i] = base[i] + add[i];
    };
    threads = std::max(1u, std::min(threads, 16u));
    if (n < (1ull << 20) || threads == 1) { body(0, n); return; }
    uint64_t per = (n + threads - 1) / threads;
    per = (per + 511) & ~511ull;  // whole pages per thread
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) {
        const uint64_t a = (uint64_t)t * per;
        if (a >= n) break;
        try {
            pool.emplace_back(body, a, std::min(n, a + per));
        } catch (const std::exception&) {  // no thread to be had: this range runs here
            body(a, std::min(n, a + per));
        }
    }
    body(0, std::min(n, per));
    for (auto& th : pool) th.join();
}

// out = the buffer (base == nullptr) or out = base + the buffer (base may be out)
static int readSumBuffer(ubdpt_context* c, uint32_t buffer, const double* base, double* out, unsigned threads) {
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    const int fence = (buffer >= 1 && c->work_pending) ? c->slot_fence[buffer - 1] : -1;
    if (fence >= 0) {
        // filled by ubdpt_enqueue_passes_separate: waits for the batches of that call only; should one of them have to
        // be rendered again, for that too
        for (hipEvent_t ev : c->fences[fence]) { HIPCK(c, hipEventSynchronize(ev)); HIPCK(c, hipStreamWaitEvent(c->stream, ev, 0)); }
        c->reapFinished();
        if (!c->redo.empty()) { int ss = settle(c); if (ss != UBDPT_SUCCESS) return ss; }
    } else {
        int fs_ = finishPending(c);
        if (fs_ != UBDPT_SUCCESS) return fs_;
    }
    launchInterleave(c->d_fb + (uint64_t)buffer * 3 * c->npix, c->npix, c->d_staging, c->stream);
    HIPCK(c, hipGetLastError());
    if (!base) {
        HIPCK(c, hipMemcpyAsync(out, c->d_staging, sizeof(double) * 3 * c->npix, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
    } else {
        // the reference-shaped renderPass comes here once per pass: a pinned staging buffer kept for the life of the
        // context (no page faults, full PCIe rate) instead of a fresh pageable one per call
        if (!c->h_staging) HIPCK(c, hipHostMalloc((void**)&c->h_staging, sizeof(double) * 3 * c->npix, hipHostMallocDefault));
        HIPCK(c, hipMemcpyAsync(c->h_staging, c->d_staging, sizeof(double) * 3 * c->npix, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        addOnto(base, c->h_staging, out, 3 * c->npix, threads);
    }
    return UBDPT_SUCCESS;
}

int ubdpt_read_sum(ubdpt_context* c, double* rgb, int add_into) {
    if (!c || !rgb) return UBDPT_INVALID_ARG;
    return readSumBuffer(c, 0, add_into ? rgb : nullptr, rgb, 1);
}

int ubdpt_read_sum_slot(ubdpt_context* c, uint32_t slot, double* rgb, int add_into) {
    if (!c || !rgb || slot >= c->pass_slots) return UBDPT_INVALID_ARG;
    return readSumBuffer(c, 1 + slot, add_into ? rgb : nullptr, rgb, 1);
}

int ubdpt_read_sum_slot_onto(ubdpt_context* c, uint32_t slot, const double* base, double* out, uint32_t host_threads) {
    if (!c || !out || slot >= c->pass_slots) return UBDPT_INVALID_ARG;
    return readSumBuffer(c, 1 + slot, base, out, host_threads);
}

// uengine/uconverter.cpp:3-76: the fit functions and their sums over the three wavelength bins
static RgbFit makeRgbFit() {
    auto xFit = [](double wave) {
        double t1 = (wave - 442.0) * ((wave < 442.0) ? 0.0624 : 0.0374);
        double t2 = (wave - 599.8) * ((wave < 599.8) ? 0.0264 : 0.0323);
        double t3 = (wave - 501.1) * ((wave < 501.1) ? 0.0490 : 0.0382);
        return 0.362 * std::exp(-0.5 * t1 * t1) + 1.056 * std::exp(-0.5 * t2 * t2) - 0.065 * std::exp(-0.5 * t3 * t3);
    };
    auto yFit = [](double wave) {
        double t1 = (wave - 568.8) * ((wave < 568.8) ? 0.0213 : 0.0247);
        double t2 = (wave - 530.9) * ((wave < 530.9) ? 0.0613 : 0.0322);
        return 0.821 * std::exp(-0.5 * t1 * t1) + 0.286 * std::exp(-0.5 * t2 * t2);
    };
    auto zFit = [](double wave) {
        double t1 = (wave - 437.0) * ((wave < 437.0) ? 0.0845 : 0.0278);
        double t2 = (wave - 459.0) * ((wave < 459.0) ? 0.0385 : 0.0725);
        return 1.217 * std::exp(-0.5 * t1 * t1) + 0.681 * std::exp(-0.5 * t2 * t2);
    };
    RgbFit f;
    f.sumX = f.sumY = f.sumZ = mk3(0, 0, 0);
    for (size_t l = 0; l < 123; l++) {
        f.sumX.x += xFit(626 + l); f.sumX.y += xFit(503 + l); f.sumX.z += xFit(380 + l);
        f.sumY.x += yFit(626 + l); f.sumY.y += yFit(503 + l); f.sumY.z += yFit(380 + l);
        f.sumZ.x += zFit(626 + l); f.sumZ.y += zFit(503 + l); f.sumZ.z += zFit(380 + l);
    }
    return f;
}

int ubdpt_image_rgb(ubdpt_context* c, uint64_t passes, double gamma, double* rgb) {
    if (!c || !rgb) return UBDPT_INVALID_ARG;
    if (passes == 0 || !(gamma > 0)) return UBDPT_NO_DATA;  // uengine/uengine.cpp:184-185
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    static const RgbFit fit = makeRgbFit();
    launchImageRgb(c->d_fb, c->npix, (double)passes, gamma, fit, c->d_staging, c->stream);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(rgb, c->d_staging, sizeof(double) * 3 * c->npix, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return UBDPT_SUCCESS;
}

int ubdpt_upload_sum(ubdpt_context* c, const double* rgb) {
    if (!c) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    if (!rgb) {
        HIPCK(c, hipMemsetAsync(c->d_fb, 0, sizeof(double) * 3 * c->npix, c->stream));
    } else {
        HIPCK(c, hipMemcpyAsync(c->d_staging, rgb, sizeof(double) * 3 * c->npix, hipMemcpyHostToDevice, c->stream));
        launchDeinterleave(c->d_staging, c->npix, c->d_fb, c->stream);
        HIPCK(c, hipGetLastError());
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return UBDPT_SUCCESS;
}

int ubdpt_intersect_batch(ubdpt_context* c, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits) {
    if (!c || (n && (!rays || !hits))) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    size_t chunk = (size_t)c->cap_paths;
    ubdpt_ray* d_rays = nullptr;
    ubdpt_hit* d_hits = nullptr;
    size_t m = n < chunk ? n : chunk;
    if (n == 0) return UBDPT_SUCCESS;
    HIPCK(c, hipMalloc((void**)&d_rays, sizeof(ubdpt_ray) * m));
    if (hipMalloc((void**)&d_hits, sizeof(ubdpt_hit) * m) != hipSuccess) { hipFree(d_rays); return UBDPT_HIP_ERROR; }
    int st = UBDPT_SUCCESS;
    for (size_t off = 0; off < n && st == UBDPT_SUCCESS; off += m) {
        size_t k = n - off < m ? n - off : m;
        hipError_t e = hipMemcpyAsync(d_rays, rays + off, sizeof(ubdpt_ray) * k, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            c->beginTimed(K_INTERSECT, (double)k);
            launchIntersectBatch(c->S, d_rays, k, d_hits, c->lanes[0].d_ovf, c->ovf_lanes, c->ovf_cap, c->d_counters + K_INTERSECT * kNumCounters, c->stream);
            c->endTimed();
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hits + off, d_hits, sizeof(ubdpt_hit) * k, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->last_hip = e; st = UBDPT_HIP_ERROR; }
    }
    hipFree(d_rays); hipFree(d_hits);
    c->collectTimed();
    if (st == UBDPT_SUCCESS) st = fetchCounters(c);
    if (st == UBDPT_SUCCESS) st = reportStackOverflows(c);
    return st;
}

int ubdpt_intersect_batch_noalloc(ubdpt_context* c, const ubdpt_ray* device_rays, size_t n, ubdpt_hit* device_hits) {
    if (!c || (n && (!device_rays || !device_hits))) return UBDPT_INVALID_ARG;
    if (n > UBDPT_NOALLOC_MAX_RAYS) {
        fprintf(stderr, "ubdpt_intersect_batch_noalloc: %zu rays exceed the limit of %llu rays per call\n", n,
                (unsigned long long)UBDPT_NOALLOC_MAX_RAYS);
        return UBDPT_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    if (c->work_pending) return UBDPT_ERROR;  // finishing enqueued passes folds statistics (host allocations): synchronize first
    if (n == 0) return UBDPT_SUCCESS;
    if (!c->noalloc_scratch || !c->h_noalloc_counters) return UBDPT_UNINITIALIZED;
    HIPCK(c, launchIntersectBatchNoAlloc(c->S, device_rays, n, device_hits, c->noalloc_scratch, c->ovf_cap,
                                         c->h_noalloc_counters, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->h_noalloc_counters[C_STACK_OVERFLOW]) {
        fprintf(stderr, "ubdpt_intersect_batch_noalloc: traversal stack overflow; the result of this call is not valid\n");
        return UBDPT_ERROR;
    }
    return UBDPT_SUCCESS;
}

int ubdpt_visibility_batch(ubdpt_context* c, const ubdpt_segment* segs, size_t n, uint8_t* visible) {
    if (!c || (n && (!segs || !visible))) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    if (n == 0) return UBDPT_SUCCESS;
    size_t chunk = (size_t)c->cap_paths;
    size_t m = n < chunk ? n : chunk;
    ubdpt_segment* d_segs = nullptr;
    uint8_t* d_vis = nullptr;
    HIPCK(c, hipMalloc((void**)&d_segs, sizeof(ubdpt_segment) * m));
    if (hipMalloc((void**)&d_vis, m) != hipSuccess) { hipFree(d_segs); return UBDPT_HIP_ERROR; }
    int st = UBDPT_SUCCESS;
    for (size_t off = 0; off < n && st == UBDPT_SUCCESS; off += m) {
        size_t k = n - off < m ? n - off : m;
        hipError_t e = hipMemcpyAsync(d_segs, segs + off, sizeof(ubdpt_segment) * k, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            c->beginTimed(K_VISIBILITY, (double)k);
            launchVisibilityBatch(c->S, d_segs, k, d_vis, c->lanes[0].d_ovf, c->ovf_lanes, c->ovf_cap, c->d_counters + K_VISIBILITY * kNumCounters, c->stream);
            c->endTimed();
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(visible + off, d_vis, k, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->last_hip = e; st = UBDPT_HIP_ERROR; }
    }
    hipFree(d_segs); hipFree(d_vis);
    c->collectTimed();
    if (st == UBDPT_SUCCESS) st = fetchCounters(c);
    if (st == UBDPT_SUCCESS) st = reportStackOverflows(c);
    return st;
}

int ubdpt_connection_probe(ubdpt_context* c, const ubdpt_path_vertex* eye, uint32_t n_eye, const ubdpt_path_vertex* light,
                           uint32_t n_light, ubdpt_connection* out, double* s0) {
    if (!c || !eye || !light || !out || n_eye == 0 || n_light == 0 || n_eye > 255 || n_light > 255) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    const size_t pairs = (size_t)n_eye * n_light;
    // one allocation: vertex input (eye, light), the two one-path stores, their count words, the outputs
    const size_t in_bytes = sizeof(ubdpt_path_vertex) * ((size_t)n_eye + n_light);
    const size_t store_bytes = sizeof(double) * kVertexDoubles * ((size_t)n_eye + n_light);
    const size_t out_bytes = sizeof(ubdpt_connection) * pairs, s0_bytes = sizeof(double) * 3 * n_eye;
    char* d = nullptr;
    HIPCK(c, hipMalloc((void**)&d, in_bytes + store_bytes + 64 + out_bytes + s0_bytes));
    ubdpt_path_vertex* d_eye = reinterpret_cast<ubdpt_path_vertex*>(d);
    ubdpt_path_vertex* d_light = d_eye + n_eye;
    PathStore E{}, L{};
    E.v = reinterpret_cast<double*>(d + in_bytes); E.n = 1; E.maxv = n_eye; E.count_stride = 1; E.dense = n_eye; E.cap = 1;
    L.v = E.v + (size_t)kVertexDoubles * n_eye; L.n = 1; L.maxv = n_light; L.count_stride = 1; L.dense = n_light; L.cap = 1;
    E.count = reinterpret_cast<uint32_t*>(d + in_bytes + store_bytes);
    L.count = E.count + 8;
    ubdpt_connection* d_out = reinterpret_cast<ubdpt_connection*>(d + in_bytes + store_bytes + 64);
    double* d_s0 = reinterpret_cast<double*>(d + in_bytes + store_bytes + 64 + out_bytes);
    hipError_t e = hipMemcpyAsync(d_eye, eye, sizeof(ubdpt_path_vertex) * n_eye, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_light, light, sizeof(ubdpt_path_vertex) * n_light, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launchConnectionProbe(c->S, d_eye, n_eye, d_light, n_light, E, L, d_out, d_s0, c->lanes[0].d_ovf, c->ovf_lanes,
                              c->ovf_cap, c->d_counters + K_VISIBILITY * kNumCounters, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && s0) e = hipMemcpyAsync(s0, d_s0, s0_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) { c->last_hip = e; return UBDPT_HIP_ERROR; }
    int st = fetchCounters(c);
    if (st == UBDPT_SUCCESS) st = reportStackOverflows(c);
    return st;
}

static int bsdfBatch(ubdpt_context* c, int mode, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) {
    if (!c || (n && (!q || !r))) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIPCK(c, hipSetDevice(c->device));
    { int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    if (n == 0) return UBDPT_SUCCESS;
    ubdpt_bsdf_query* dq = nullptr;
    ubdpt_bsdf_result* dr = nullptr;
    HIPCK(c, hipMalloc((void**)&dq, sizeof(*dq) * n));
    if (hipMalloc((void**)&dr, sizeof(*dr) * n) != hipSuccess) { hipFree(dq); return UBDPT_HIP_ERROR; }
    hipError_t e = hipMemcpyAsync(dq, q, sizeof(*dq) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launchBsdfBatch(mode, dq, n, dr, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(r, dr, sizeof(*dr) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dq); hipFree(dr);
    if (e != hipSuccess) { c->last_hip = e; return UBDPT_HIP_ERROR; }
    return UBDPT_SUCCESS;
}
int ubdpt_bsdf_eval_batch(ubdpt_context* c, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) { return bsdfBatch(c, 0, q, n, r); }
int ubdpt_bsdf_pdf_batch(ubdpt_context* c, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) { return bsdfBatch(c, 1, q, n, r); }
int ubdpt_bsdf_scatter_batch(ubdpt_context* c, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) { return bsdfBatch(c, 2, q, n, r); }

int ubdpt_get_object_info(ubdpt_context* c, uint32_t object, double invW[16], double* area, double* probability,
                          double bounding_sphere[16]) {
    if (!c || object >= c->h_objects.size()) return UBDPT_INVALID_ARG;
    const DObject& o = c->h_objects[object];
    if (invW) memcpy(invW, &o.invW, sizeof(double) * 16);
    if (area) *area = o.area;
    if (probability) *probability = o.prob;
    if (bounding_sphere) memcpy(bounding_sphere, &o.bsphere, sizeof(double) * 16);
    return UBDPT_SUCCESS;
}

int ubdpt_get_camera_info(ubdpt_context* c, double view[16], double inv_view[16], double* image_plane_distance) {
    if (!c) return UBDPT_INVALID_ARG;
    if (view) memcpy(view, &c->S.V, sizeof(double) * 16);
    if (inv_view) memcpy(inv_view, &c->S.invV, sizeof(double) * 16);
    if (image_plane_distance) *image_plane_distance = c->S.image_plane_distance;
    return UBDPT_SUCCESS;
}

int ubdpt_get_image_size(ubdpt_context* c, uint64_t* width, uint64_t* height) {
    if (!c) return UBDPT_INVALID_ARG;
    if (width) *width = c->params.img_res_x;
    if (height) *height = c->params.img_res_y;
    return UBDPT_SUCCESS;
}

int ubdpt_get_batch_info(ubdpt_context* c, uint64_t* pixel_samples_per_batch, uint32_t* lanes, uint32_t* wavefront_depth) {
    if (!c) return UBDPT_INVALID_ARG;
    if (pixel_samples_per_batch) *pixel_samples_per_batch = c->cap_paths;
    if (lanes) *lanes = (uint32_t)c->lanes.size();
    if (wavefront_depth) *wavefront_depth = c->wf_depth;
    return UBDPT_SUCCESS;
}

int ubdpt_get_recovery_info(ubdpt_context* c, uint64_t* batches_rendered_again, uint64_t* pixel_samples_per_batch_now) {
    if (!c) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->work_pending) { hipSetDevice(c->device); int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    if (batches_rendered_again) *batches_rendered_again = c->batches_redone;
    if (pixel_samples_per_batch_now) *pixel_samples_per_batch_now = std::min(c->cap_paths, c->batch_limit);
    return UBDPT_SUCCESS;
}

int ubdpt_get_stats(ubdpt_context* c, ubdpt_stats* out) {
    if (!c || !out) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->work_pending) { hipSetDevice(c->device); int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    *out = c->stats;
    out->num_kernel_stats = 0;
    for (int k = 0; k < K_NUM && out->num_kernel_stats < UBDPT_MAX_KERNEL_STATS; k++) {
        if (c->kernel_launches[k] == 0) continue;
        ubdpt_kernel_stat& ks = out->kernels[out->num_kernel_stats++];
        memset(&ks, 0, sizeof(ks));
        snprintf(ks.name, sizeof(ks.name), "%s", kKernelNames[k]);
        ks.launches = c->kernel_launches[k]; ks.ms = c->kernel_ms[k]; ks.units = c->kernel_units[k];
        const unsigned long long* h = c->kernel_counters[k];
        ks.closest_hit_rays = h[C_CLOSEST]; ks.shadow_rays = h[C_SHADOW]; ks.nodes_visited = h[C_NODES]; ks.prims_tested = h[C_PRIMS];
        ks.wave_node_steps = h[C_WAVE_NODES]; ks.wave_prim_steps = h[C_WAVE_PRIMS]; ks.max_stack_entries = h[C_MAX_STACK];
        ks.vertices = h[C_EYE_V] + h[C_LIGHT_V]; ks.fb_updates = h[C_FB];
    }
    return UBDPT_SUCCESS;
}

int ubdpt_reset_stats(ubdpt_context* c) {
    if (!c) return UBDPT_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->work_pending) { hipSetDevice(c->device); int fs_ = finishPending(c); if (fs_ != UBDPT_SUCCESS) return fs_; }
    double build = c->stats.build_seconds;
    memset(&c->stats, 0, sizeof(c->stats));
    c->stats.build_seconds = build;
    for (int k = 0; k < K_NUM; k++) { c->kernel_ms[k] = 0; c->kernel_units[k] = 0; c->kernel_launches[k] = 0; }
    memset(c->kernel_counters, 0, sizeof(c->kernel_counters));
    return UBDPT_SUCCESS;
}

}  // extern "C"
