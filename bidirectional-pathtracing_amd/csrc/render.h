// render.h — host-visible launch interface of the device code (render.hip, bvh.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/ubdpt.h"
#include "paths.h"

namespace ub {

struct RenderArgs {
    DScene S;
    PathStore E, L;        // eye / light subpath vertices of the batch
    double* fb;            // planar sum buffers of the batch lane (R[npix], G[npix], B[npix] each): what the batch adds is
                           // collected here and handed on to the context's buffers by k_commit_batch when the batch is complete
    uint64_t first_pass, pass_stride, seed;
    uint64_t path_offset;  // index of the batch's first pixel-sample among the call's n_passes * npix
    uint64_t fb_slots;     // 0: everything adds into the first buffer at fb; > 0: the passes are kept apart, the q-th pass the
                           // batch touches adds into buffer q
    uint64_t batch_pass0;  // path_offset / npix and path_offset % npix, split on the host: the kernels then divide
    uint64_t batch_pixel0; // (batch_pixel0 + i), a number that almost always fits 32 bits, instead of a 64-bit index
    int32_t* ovf;          // traversal-stack overflow, [kOverflowStack][ovf_stride]
    uint64_t ovf_stride;
    uint32_t ovf_cap;      // entries of a lane's whole stack, on-chip part included (ubdpt_options.traversal_stack_entries)
    uint32_t pad_ovf;
    unsigned long long* counters;
    uint32_t* batch_fail;  // set when the batch cannot be completed (the pool of deep vertex slots ran empty): its sums are dropped
    uint64_t* pairs;                  // connection queue: (path << 16) | (s << 8) | t
    unsigned long long pair_capacity;
    unsigned long long* pair_count;   // device counter, zeroed before every batch
    // wavefront subpath generation (wavefront.h); subpath id in [0, 2n)
    double* carry;           // [2n][kCarryDoubles]: the ray in flight and what the next bounce needs of the last two vertices
    double* wf_hit;          // [2n][kHitDoubles]: d, u, v, (object | face << 32) of the closest hit found by k_wf_trace
    uint64_t* vis_pairs;     // connection pairs that passed the shadow ray
    uint32_t* vis_count;
    double* rays;            // shadow-ray queue, kRayDoubles per ray: origin, direction, length
    uint64_t* ray_pairs;     // the pair each ray of the queue belongs to
    uint32_t* ray_count;     // rays in the queue; ray_count[1] = the position the persistent waves claim chunks at; ray_count[2] = pixel-samples k_pair_enum could not queue; ray_count[3] = pairs k_pair_eval put aside for its second launch (in ray_pairs)
    uint32_t* ray_head;
    uint8_t* ray_flag;       // one per ray: 1 = nothing blocks it
    uint32_t* wf_trace_q;    // [2][2n] ping-pong queues of live subpaths
    uint32_t* wf_count;      // [depth][16]: [d][0] length of the depth's trace queue, [d][3] the position the persistent waves claim chunks at; zeroed before every batch
    // The count word of every eye / light subpath once more, written when the subpath ends, 4 bytes apart: the pair
    // enumeration reads two words per pixel-sample, and out of the carry records each of them costs a whole line.
    uint32_t* fin_E;
    uint32_t* fin_L;
};

constexpr uint32_t kMaxWfDepth = 30;  // most bounces that can run as wavefront launches before the tail loop takes over
constexpr int kWfQStride = 16;        // counters per depth
constexpr int kRayDoubles = 8;        // one 64-byte line per shadow ray
constexpr int kCarryDoubles = 16;     // two 64-byte lines per subpath
constexpr int kCarryCountSlot = 7;    // the double of the first line that holds the subpath's count word
constexpr int kHitDoubles = 4;        // half a line per subpath
void launchWfGen(const RenderArgs& A, bool stats, hipStream_t s);
void launchWfTail(const RenderArgs& A, uint32_t depth, unsigned blocks, bool stats, hipStream_t s);
void launchWfTrace(const RenderArgs& A, uint32_t depth, unsigned blocks, unsigned small_scene_blocks, bool stats, hipStream_t s);
void launchPairVisibleLds(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s);  // scenes whose tree sits in LDS: set-up + rays + compaction
void launchWfShade(const RenderArgs& A, uint32_t depth, unsigned blocks, bool stats, hipStream_t s);
int sceneLdsLevel(const DScene& S);  // 0 = scene in HBM, 1 = object table in LDS, 2 = tree and primitives too
void launchPairSetup(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s);
void launchRaysVisible(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s);
void launchPairCompact(const RenderArgs& A, unsigned blocks, hipStream_t s);
void launchPairEval(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s);

constexpr int kNumCounters = 16;
// counters, indices into RenderArgs::counters (one block of kNumCounters words per kernel); C_MAX_STACK is a maximum, the others are sums
enum { C_CLOSEST = 0, C_SHADOW, C_EYE_V, C_LIGHT_V, C_CONN, C_SPLAT, C_NODES, C_PRIMS, C_CAPPED, C_STACK_OVERFLOW, C_FB,
       C_WAVE_NODES, C_WAVE_PRIMS, C_MAX_STACK, C_POOL_OVERFLOW, C_NUM };
static_assert(C_NUM <= kNumCounters, "counter block too small");

void launchEye(const RenderArgs& A, bool stats, hipStream_t s);
void launchLight(const RenderArgs& A, bool stats, hipStream_t s);
void launchConnect(const RenderArgs& A, bool stats, hipStream_t s);
void launchPairEnum(const RenderArgs& A, unsigned blocks, bool stats, hipStream_t s);
void launchIntersectBatch(const DScene& S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits, int32_t* ovf,
                          uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s);
void launchVisibilityBatch(const DScene& S, const ubdpt_segment* segs, size_t n, uint8_t* vis, int32_t* ovf,
                           uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s);
// The entry point that allocates nothing: a fixed number of lanes walk the ray batch with a stride; their
// traversal-stack overflow and the counters live in a static device array (render.hip).
constexpr uint64_t kNoAllocLanes = 32768;
hipError_t noAllocScratch(int device, void** out);  // device must be current; resolves the array's address once per device
hipError_t launchIntersectBatchNoAlloc(const DScene& S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits, void* scratch_ptr,
                                       uint32_t ovf_cap, unsigned long long* host_counters, hipStream_t s);
// stores E / L: one path (n = 1), maxv >= the vertex counts, dense count words
void launchConnectionProbe(const DScene& S, const ubdpt_path_vertex* eye, uint32_t n_eye, const ubdpt_path_vertex* light,
                           uint32_t n_light, PathStore E, PathStore L, ubdpt_connection* out, double* s0, int32_t* ovf,
                           uint64_t ovf_stride, uint32_t ovf_cap, unsigned long long* counters, hipStream_t s);
void launchBsdfBatch(int mode, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r, hipStream_t s);
void launchInterleave(const double* fb, uint64_t npix, double* out, hipStream_t s);
void launchInterleaveAdd(const double* fb, uint64_t npix, double* out, hipStream_t s);
void launchDeinterleave(const double* in, uint64_t npix, double* fb, hipStream_t s);
// End of a batch: the lane's `slots` sum buffers (3 * npix doubles each) are added to the buffers from `target` on and
// cleared, its counter words are added to `counters` and cleared; when *fail is set the sums and the counters are dropped
// (all but C_POOL_OVERFLOW), and the host renders the batch again in smaller pieces.
void launchCommitBatch(double* lane_fb, uint64_t npix, uint32_t slots, double* target, const uint32_t* fail,
                       unsigned long long* lane_counters, unsigned long long* counters, uint32_t counter_words, hipStream_t s);
// sums of the CIE 1931 fits over the three 123 nm bins (uengine/uconverter.cpp:31-76), computed once on the host
struct RgbFit { d3 sumX, sumY, sumZ; };
void launchImageRgb(const double* fb, uint64_t npix, double passes, double gamma, const RgbFit& fit, double* out,
                    hipStream_t s);

// LBVH build (bvh.hip). Inputs are device arrays; returns hipSuccess or the failing HIP status.
// `prims_unsorted` holds one Prim per primitive in object order; on return `prims_out` holds them in
// tree order and `nodes_out` the tree (max(1, n-1) nodes, root = node 0).
struct BvhBuildResult {
    BvhNode* nodes;   // device, owned by caller after return
    ExactNode* xnodes; // device, owned by caller after return; null above kExactTreePrims primitives
    Prim* prims;      // device, owned by caller after return
    uint32_t num_nodes;
    float scene_lo[3], scene_hi[3];
};
hipError_t buildLbvh(const DObject* d_objects, const Prim* d_prims_unsorted, uint32_t num_prims, hipStream_t s,
                     BvhBuildResult* out);

}  // namespace ub
