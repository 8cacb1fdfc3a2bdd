/*
 * ubdpt.h — C ABI of the MI355X bidirectional-path-tracing backend.
 *
 * This is the drop-in boundary for the reference's renderer plug-in point
 *   class URenderer { initialize / renderPass / stop }   (uengine/urenderer.h:10-16)
 * as it is driven by UEngine                              (uengine/uengine.cpp:33-63,144-180,204-208).
 * A URenderer subclass living in the reference tree forwards to these entry points:
 * initialize -> ubdpt_create, renderPass -> ubdpt_render_passes + ubdpt_read_sum,
 * stop -> ubdpt_request_stop. Only plain pointers, sizes and POD structs cross
 * the boundary; every function returns an int status and never throws.
 *
 * Because UScene exposes geometry only through opaque virtuals
 * (uengine/uscene.h:16-18, uengine/uobject.h:14-18) the scene crosses the boundary as a
 * flat description that carries exactly the data the reference's uapp classes hold:
 *   Object   {W, material, model}            uapp/object.h:10-24
 *   Emitter  {power}                         uapp/emitter.cpp:3-7
 *   Mesh     {faces of 3 x MeshVertex}       uapp/mesh.h:13-22  (floats, as assimp delivers them, uapp/mesh.cpp:20-30)
 *   Material {LatexPaint, Glossy(d,s), PerfectMirror, Dielectric(eta)}   uapp/material.h:17-104
 *   Texture  {TextureColor, TextureImg}      uapp/texturecolor.cpp:3-11, uapp/textureimg.cpp:19-33
 *   UCamera  {ratio, vfov, pos, lookAt}      uengine/ucamera.cpp:3-14
 * Everything derived (inverse matrices, areas, emitter pick probabilities, mesh bounding
 * spheres, face pick probabilities, view matrix, acceleration structure) is computed by
 * the library, following the reference.
 */
#ifndef UBDPT_H
#define UBDPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: superset of UResult (uengine/uengine.h:18), same order for the shared part ---- */
enum {
    UBDPT_SUCCESS        = 0, /* UResult::USuccess       */
    UBDPT_UNINITIALIZED  = 1, /* UResult::UUninitialized */
    UBDPT_INVALID_SCENE  = 2, /* UResult::UInvalidScene  */
    UBDPT_INVALID_FORMAT = 3, /* UResult::UInvalidFormat */
    UBDPT_STOPPED        = 4, /* UResult::UStopped       */
    UBDPT_NO_DATA        = 5, /* UResult::UNoData        */
    UBDPT_ERROR          = 6, /* UResult::UError         */
    UBDPT_INVALID_ARG    = 7,
    UBDPT_NO_DEVICE      = 8,
    UBDPT_HIP_ERROR      = 9
};

/* ---- scene description ---- */
enum { UBDPT_MODEL_SPHERE = 0, UBDPT_MODEL_MESH = 1 };
/* material kinds, uapp/material.h */
enum { UBDPT_MAT_LATEX_PAINT = 0, UBDPT_MAT_GLOSSY = 1, UBDPT_MAT_PERFECT_MIRROR = 2, UBDPT_MAT_DIELECTRIC = 3 };
enum { UBDPT_TEX_COLOR = 0, UBDPT_TEX_IMAGE = 1 };
/* BSDF lobe ids as stored in hit records and path vertices (uengine/ubsdf*.h); -1 = no lobe (absorbed) */
enum { UBDPT_LOBE_NONE = -1, UBDPT_LOBE_LAMBERTIAN = 0, UBDPT_LOBE_MIRROR = 1, UBDPT_LOBE_DIELECTRIC = 2 };

/* UCamera constructor arguments (uengine/ucamera.cpp:3-14); up vector is always (0,1,0). */
typedef struct ubdpt_camera {
    double pos[3];
    double look_at[3];
    double vfov;   /* vertical field of view, radians */
    double ratio;  /* width / height */
} ubdpt_camera;

/* Same members, order and size (56 bytes on LP64) as URenderParameters (uengine/uutils.h:11-20). */
typedef struct ubdpt_render_params {
    uint64_t img_res_x;
    uint64_t img_res_y;
    uint64_t pixel_subdiv;
    uint64_t lens_subdiv;
    uint64_t min_depth;
    double   focus_plane_distance;
    double   lens_size;
} ubdpt_render_params;

/* One corner of a mesh face: the float data aiMesh holds (uapp/mesh.cpp:14-32). 44 bytes. */
typedef struct ubdpt_mesh_vertex {
    float pos[3];
    float normal[3];
    float tangent[3];
    float uv[2];
} ubdpt_mesh_vertex;

typedef struct ubdpt_texture {
    uint32_t kind;          /* UBDPT_TEX_* */
    uint32_t width, height; /* image only */
    uint32_t reserved;
    double   color[3];      /* TextureColor value */
    uint64_t texel_offset;  /* byte offset of this image's RGBA8888 rows inside scene.texels */
} ubdpt_texture;

typedef struct ubdpt_object {
    uint32_t kind;        /* UBDPT_MODEL_* */
    uint32_t material;    /* UBDPT_MAT_* */
    double   W[16];       /* object-to-world, column-major like glm::dmat4x4 */
    double   mat_a;       /* Glossy: d;  Dielectric: eta */
    double   mat_b;       /* Glossy: s */
    uint32_t texture;     /* index into scene.textures */
    uint32_t is_emitter;  /* non-zero: uapp Emitter (object AND emitter, uapp/scene.cpp:232-237) */
    double   power[3];    /* emitter power */
    uint64_t first_face;  /* mesh: first face in scene.vertices (3 vertices per face) */
    uint64_t num_faces;   /* mesh: number of faces */
} ubdpt_object;

/* Object order is the order of UScene::objects(); emitters keep their relative order
 * (UScene::emitters()), both matter for tie-breaks and emitter selection. */
typedef struct ubdpt_scene_desc {
    ubdpt_camera             camera;
    uint32_t                 num_objects;
    uint32_t                 num_textures;
    const ubdpt_object*      objects;
    const ubdpt_texture*     textures;
    uint64_t                 num_faces;     /* total faces of all meshes */
    const ubdpt_mesh_vertex* vertices;      /* 3 * num_faces */
    uint64_t                 num_texel_bytes;
    const uint8_t*           texels;        /* RGBA8888 */
} ubdpt_scene_desc;

/* Backend knobs that have no counterpart in the reference. Zero-initialise for defaults. */
typedef struct ubdpt_options {
    uint32_t max_path_vertices; /* cap on vertices per subpath (reference: unbounded); 0 -> 32 */
    uint32_t passes_in_flight;  /* passes rendered concurrently by one wavefront batch; 0 -> auto */
    uint32_t accel;             /* 0 = LBVH, 1 = linear scan over all primitives (reference order) */
    uint32_t collect_stats;     /* non-zero: count rays / vertices / traversal steps (slower) */
    uint32_t pipeline;          /* 0 = wavefront pipeline (default); 1 = one lane per pixel-sample in the reference's
                                   loop structure, kept as a cross-check of the default */
    uint32_t wavefront_depth;   /* bounces run as wavefront launches before the per-lane tail loop; 0 -> default, max 30 */
    uint32_t streams;           /* batches in flight on separate HIP streams; 0 -> default, max 16 */
    uint32_t batch_pixel_samples; /* non-zero and below one pass: a batch is this many pixels of one pass
                                     (overrides passes_in_flight); small batches keep the path vertices cache-resident */
    uint32_t pass_slots;        /* per-pass sum buffers held on the device besides the sum buffer (0: none, max 64);
                                   ubdpt_render_passes_separate keeps up to that many passes of a call apart */
    uint32_t traversal_stack_entries; /* entries of a ray's traversal stack; 0 -> 112 (the maximum: 16 on chip, 96 in
                                   device memory). A ray that needs more makes the call that traced it return UBDPT_ERROR:
                                   a dropped subtree could hide a hit, so such a result is never handed out silently */
    uint32_t dense_vertex_slots; /* vertex slots stored for EVERY subpath of a batch; 0 -> 16 (rounded up to a multiple of 8,
                                   at most max_path_vertices). The slots beyond them are handed out in pages of 8 from a
                                   pool, to the few subpaths that get that far (1 in 10 000 reaches slot 16 on box scenes
                                   with glass): a third less memory per pixel-sample, larger batches. */
    uint32_t vertex_pool_pages;  /* size of that pool: pages per 1024 subpaths of a batch; 0 -> 128 (one subpath in eight
                                   can hold a page). A batch that needs more is not handed out: its sums are dropped on the
                                   device, it is rendered again in pieces the pool can serve (same samples, same image), and
                                   the batches after it are made that small (ubdpt_get_recovery_info tells). A scene that
                                   does this often (a closed room of mirrors) runs faster with a larger pool or with every
                                   slot stored densely (dense_vertex_slots = max_path_vertices). */
} ubdpt_options;

/* ---- batch intersection (parity probe for UScene::intersectionPoint / visibility) ---- */
typedef struct ubdpt_ray {      /* URay: origin (w=1) and direction (w=0), uengine/ugeometry.h:35-37 */
    double origin[3];
    double dir[3];
} ubdpt_ray;

typedef struct ubdpt_hit {      /* the fields UScene::intersectionPoint fills in USurfacePoint (uengine/uutils.h:22-35) */
    int32_t  hit;               /* 0 / 1 */
    int32_t  object;            /* index in scene.objects, -1 if none */
    int64_t  face;              /* face index inside the object's mesh, -1 for spheres / miss */
    double   d;                 /* ray parameter of the hit (object-local == world, uengine/ugeometry.cpp:55-58) */
    double   pos[3];            /* object-local, like USurfacePoint before renderer transforms it */
    double   Ng[3], Ns[3], Ts[3], Bs[3];
    double   tex_u, tex_v;
} ubdpt_hit;

typedef struct ubdpt_segment {  /* UScene::visibility(p0, p1) arguments */
    double p0[3];
    double p1[3];
} ubdpt_segment;

/* ---- statistics ---- */
#define UBDPT_MAX_KERNEL_STATS 24
typedef struct ubdpt_kernel_stat {
    char     name[40];
    uint64_t launches;
    double   ms;        /* HIP-event time on the backend's own stream, summed over launches */
    double   units;     /* work items summed over launches (rays, vertices, pixel-samples ...) */
    /* work counted inside this kernel (only with collect_stats; zero otherwise) */
    uint64_t closest_hit_rays, shadow_rays, nodes_visited, prims_tested, vertices, fb_updates;
    /* wave-level executions of the node step / the primitive test: nodes_visited / (64 * wave_node_steps)
     * is the SIMD utilisation of the traversal */
    uint64_t wave_node_steps, wave_prim_steps;
    uint64_t max_stack_entries;  /* deepest traversal stack any ray of this kernel needed (collect_stats / probe kernels) */
} ubdpt_kernel_stat;

typedef struct ubdpt_stats {
    uint64_t passes;            /* passes rendered since create / reset */
    uint64_t pixel_samples;     /* renderPixel invocations = paths */
    uint64_t closest_hit_rays;  /* UScene::intersectionPoint calls (only with collect_stats) */
    uint64_t shadow_rays;       /* UScene::visibility calls actually traced (only with collect_stats) */
    uint64_t eye_vertices;      /* stored eye-subpath vertices incl. lens vertex (collect_stats) */
    uint64_t light_vertices;    /* stored light-subpath vertices incl. emitter vertex (collect_stats) */
    uint64_t connections;       /* (s,t) pairs that contributed (collect_stats) */
    uint64_t splats;            /* t==1 atomic splats (collect_stats) */
    uint64_t bvh_nodes_visited; /* traversal steps (collect_stats) */
    uint64_t prims_tested;      /* primitive tests (collect_stats) */
    uint64_t capped_subpaths;   /* subpaths that hit max_path_vertices (collect_stats) */
    double   render_seconds;    /* host wall time inside ubdpt_render_passes, incl. final sync */
    double   build_seconds;     /* scene upload + acceleration structure build */
    uint32_t num_kernel_stats;
    uint32_t reserved;
    ubdpt_kernel_stat kernels[UBDPT_MAX_KERNEL_STATS];
} ubdpt_stats;

typedef struct ubdpt_context ubdpt_context;

/* Library / device probes. */
int         ubdpt_version(void);
const char* ubdpt_status_string(int status);
int         ubdpt_device_count(int* count);

/*
 * URenderer::initialize (uengine/ubdptrenderer.cpp:7-33) + UEngine::setScene's
 * computeEmitterProbabilities (uengine/uengine.cpp:15-20, uengine/uscene.cpp:4-19) +
 * UEngine::initPixelBuffers (uengine/uengine.cpp:22-31): flattens the scene onto the device,
 * builds the acceleration structure and allocates the zeroed W x H sum buffer.
 * `options` may be NULL.
 */
int ubdpt_create(const ubdpt_scene_desc* scene, const ubdpt_render_params* params,
                 const ubdpt_options* options, int device, ubdpt_context** out);

/* Releases everything ubdpt_create allocated (UEngine drops its unique_ptr<URenderer>, uengine/uengine.h:62). */
int ubdpt_destroy(ubdpt_context* ctx);

/*
 * URenderer::renderPass (uengine/ubdptrenderer.cpp:35-81) for passes
 * first_pass, first_pass + pass_stride, ... (n_passes of them): every pass adds one
 * renderPixel sample (uengine/ubdptrenderer.cpp:88-154) per pixel into the device sum buffer,
 * strata chosen by the pass index (:95-96). pass_stride > 1 is the multi-GPU partition.
 * Returns UBDPT_STOPPED if ubdpt_request_stop arrived; passes already finished stay accumulated
 * and *passes_done (may be NULL) tells how many.
 */
int ubdpt_render_passes(ubdpt_context* ctx, uint64_t first_pass, uint32_t n_passes,
                        uint64_t pass_stride, uint64_t seed, uint32_t* passes_done);

/*
 * UEngine::imageRGB (uengine/uengine.cpp:182-202) on the device sum buffer: radiance = sum / passes,
 * then UConverter::radianceToRGB (uengine/uconverter.cpp:78-126: CIE XYZ fit, sRGB matrix, gamma).
 * rgb receives H x W x 3 doubles in [0,1], row-major like the sum buffer. Returns UBDPT_NO_DATA when
 * passes == 0 or gamma <= 0, as the reference does. Only W*H*3 display values cross PCIe.
 */
int ubdpt_image_rgb(ubdpt_context* ctx, uint64_t passes, double gamma, double* rgb);

/*
 * The same work as ubdpt_render_passes, enqueued without waiting for the GPU: consecutive calls keep
 * the device busy across call boundaries (the reference's counterpart is UEngine calling
 * renderPass again as soon as the previous pass returned, uapp/appmanager.cpp:486).
 * Every entry point that reads results (read_sum*, get_stats, ...) waits for the enqueued passes
 * first; ubdpt_synchronize waits explicitly and reports errors of the enqueued work.
 * The passes of a call that do not fill a batch are started together with the next call's passes when that call
 * continues where this one ended (same seed and stride, next pass), otherwise before whatever comes next; they are
 * counted in *passes_enqueued of the call that took them.
 */
int ubdpt_enqueue_passes(ubdpt_context* ctx, uint64_t first_pass, uint32_t n_passes,
                         uint64_t pass_stride, uint64_t seed, uint32_t* passes_enqueued);
int ubdpt_synchronize(ubdpt_context* ctx);

/*
 * The update_progress_callback of URenderer::renderPass (uengine/ubdptrenderer.cpp:54-62 counts finished pixels of
 * the pass against W * H): pixel-samples handed to the GPU since ubdpt_create and how many of them the GPU has
 * finished, without waiting for anything. A caller that enqueues passes polls it to drive a progress display.
 * Either pointer may be NULL.
 */
int ubdpt_query_progress(ubdpt_context* ctx, uint64_t* samples_enqueued, uint64_t* samples_finished);

/* URenderer::stop (uengine/ubdptrenderer.cpp:83-86): callable from any thread. */
int ubdpt_request_stop(ubdpt_context* ctx);

/*
 * The sum buffer UEngine hands to renderPass (uengine/uengine.cpp:153-164): H x W x 3 doubles,
 * row-major [y][x][rgb] exactly like UBuffer2D<glm::dvec3> (uengine/uutils.h:37-98).
 * add_into != 0: rgb[i] += device sum (the adapter's "add this pass to the engine's buffer");
 * add_into == 0: rgb[i]  = device sum.
 */
int ubdpt_read_sum(ubdpt_context* ctx, double* rgb, int add_into);
/* Same data, but written to DEVICE memory owned by the caller (e.g. a buffer handed to an RCCL reduce). */
int ubdpt_read_sum_device(ubdpt_context* ctx, double* device_rgb);
/* device_rgb[i] += device sum: the step before the inter-GPU reduce when several contexts share one device (the
 * reference's counterpart is the += of every thread into the one pixel buffer, uengine/ubdptrenderer.cpp:144,153). */
int ubdpt_accumulate_sum_device(ubdpt_context* ctx, double* device_rgb);
/* Replaces the device sum (resume from a .ur file, uengine/uengine.cpp:107-116). NULL clears it. */
int ubdpt_upload_sum(ubdpt_context* ctx, const double* rgb);

/*
 * For a caller that must hand back ONE pass at a time, as URenderer::renderPass does
 * (uengine/uengine.cpp:153-164: the engine owns the running sum and adds one pass per call), but
 * wants the device to work on batches of passes: the n_passes <= ubdpt_options.pass_slots passes of
 * the call are rendered together, the pass with index j among them adding into per-pass buffer j
 * (cleared first) instead of the device sum, which this call leaves alone. ubdpt_read_sum_slot then
 * delivers the passes one by one. The samples of a pass do not depend on what it is batched with.
 */
int ubdpt_render_passes_separate(ubdpt_context* ctx, uint64_t first_pass, uint32_t n_passes,
                                 uint64_t pass_stride, uint64_t seed, uint32_t* passes_done);
int ubdpt_read_sum_slot(ubdpt_context* ctx, uint32_t slot, double* rgb, int add_into);
/*
 * out[i] = base[i] + per-pass buffer `slot`, the add running on up to host_threads CPU threads (1..16; the count
 * URenderer::renderPass is given, uengine/uengine.cpp:158-164). With base = the engine's read buffer and out = its
 * write buffer this is UEngine::renderPass's copy of the finished sum (uengine/uengine.cpp:153-156) and the renderer's
 * += of one pass (uengine/ubdptrenderer.cpp:144,153) in one sweep over the image; base may be out (the += alone);
 * base = NULL delivers the pass itself.
 */
int ubdpt_read_sum_slot_onto(ubdpt_context* ctx, uint32_t slot, const double* base, double* out, uint32_t host_threads);
/*
 * The same, enqueued without waiting for the GPU, into per-pass buffers first_slot .. first_slot + n_passes - 1
 * (first_slot + n_passes <= pass_slots), so that the device renders the next window of passes into one half of
 * the buffers while the host consumes the other half (the reference's counterpart: the render loop on its own thread,
 * uapp/appmanager.cpp:430 and :486, while the GUI thread converts the finished buffer, :104). ubdpt_read_sum_slot of a
 * buffer filled by this call waits for the passes of THAT call only; errors of enqueued work are reported by
 * ubdpt_synchronize. Re-using buffers whose passes were never read is allowed (the clear waits for them).
 */
int ubdpt_enqueue_passes_separate(ubdpt_context* ctx, uint64_t first_pass, uint32_t n_passes,
                                  uint64_t pass_stride, uint64_t seed, uint32_t first_slot,
                                  uint32_t* passes_enqueued);

/* UScene::intersectionPoint (uengine/uscene.cpp:41-66) for n rays, through the backend's accelerator. */
int ubdpt_intersect_batch(ubdpt_context* ctx, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits);
/*
 * No counterpart in the reference. For callers that cannot allocate device memory: tight serving loops, harnesses
 * that forbid hipMalloc, code that runs between the captures of HIP graphs (the call itself waits for its kernel and
 * reads a counter back, so it cannot be part of a capture). The same query as ubdpt_intersect_batch, but rays and hits
 * are DEVICE buffers of the caller, and what the kernel needs besides them (the off-chip part of the traversal
 * stacks, the overflow counter) lives in a fixed array inside the library (12.6 MB per GPU, held until the process
 * ends) instead of in allocated memory: the call makes no hipMalloc / hipFree and allocates nothing on the host.
 * n may not exceed UBDPT_NOALLOC_MAX_RAYS (UBDPT_INVALID_ARG otherwise). Passes enqueued on the context must have
 * been waited for (ubdpt_synchronize) before the call, UBDPT_ERROR otherwise; UBDPT_ERROR as well when a traversal stack
 * overflowed. Every call on one GPU uses the same array, whatever the context: calls that overlap on one GPU race.
 * The call returns when the hits are complete.
 */
#define UBDPT_NOALLOC_MAX_RAYS (1ull << 20)
int ubdpt_intersect_batch_noalloc(ubdpt_context* ctx, const ubdpt_ray* device_rays, size_t n, ubdpt_hit* device_hits);
/* UScene::visibility (uengine/uscene.cpp:21-39) for n point pairs: visible[i] = 1 / 0. */
int ubdpt_visibility_batch(ubdpt_context* ctx, const ubdpt_segment* segs, size_t n, uint8_t* visible);

/*
 * Derived per-object data, for checking the flattening against the reference's constructors:
 * invW (uapp/object.cpp:8), area (uapp/emitter.cpp:6), pick probability (uengine/uscene.cpp:4-19),
 * bounding-sphere matrix (uapp/mesh.cpp:38-86; identity for spheres). Any out pointer may be NULL.
 */
int ubdpt_get_object_info(ubdpt_context* ctx, uint32_t object, double invW[16], double* area,
                          double* probability, double bounding_sphere[16]);
/* View matrix and image-plane distance the renderer snapshots (uengine/ubdptrenderer.cpp:11-14). */
int ubdpt_get_camera_info(ubdpt_context* ctx, double view[16], double inv_view[16], double* image_plane_distance);

/* img_res_x / img_res_y the context was created with (URenderParameters, uengine/uutils.h:11-20): the size of the sum buffer. */
int ubdpt_get_image_size(ubdpt_context* ctx, uint64_t* width, uint64_t* height);
/* How the context batches its work (no counterpart in the reference, whose unit of work is one pixel,
 * uengine/ubdptrenderer.cpp:43-75): pixel-samples of one batch, batches in flight side by side (lanes), bounces that
 * run as wavefront launches before the tail loop. What ubdpt_create settled on after sizing the batches to the memory
 * that was free; the measurement tools relate per-launch counters to pixel-samples with it. Any pointer may be NULL. */
int ubdpt_get_batch_info(ubdpt_context* ctx, uint64_t* pixel_samples_per_batch, uint32_t* lanes, uint32_t* wavefront_depth);

/* Batches that exhausted the pool of deep vertex slots (ubdpt_options.vertex_pool_pages) and were rendered again in
 * smaller pieces, since the context was created, and the pixel-samples a batch holds now (lowered whenever that happened).
 * No counterpart in the reference, whose subpaths are std::vectors that grow (uengine/ubdptrenderer.cpp:224-229).
 * Waits for the enqueued passes. Any pointer may be NULL. */
int ubdpt_get_recovery_info(ubdpt_context* ctx, uint64_t* batches_rendered_again, uint64_t* pixel_samples_per_batch_now);

int ubdpt_get_stats(ubdpt_context* ctx, ubdpt_stats* out);
int ubdpt_reset_stats(ubdpt_context* ctx);

/*
 * BSDF probes (UBsdf::samplePSA / pPSA / scatter, uengine/ubsdf.h:22-38), evaluated on the device
 * for n independent queries. lobe = UBDPT_LOBE_*; eta is used by the dielectric; tex is the texture value.
 * frame = Ng, Ns, Ts, Bs (4 x 3 doubles). For scatter, u1/u2 are the uniform numbers the reference
 * would draw, in its draw order.
 */
typedef struct ubdpt_bsdf_query {
    int32_t lobe;
    int32_t reserved;
    double  eta;
    double  tex[3];
    double  frame[12];
    double  wi[3];     /* samplePSA: wiT; pPSA: wsT; scatter: w (same space as frame) */
    double  wo[3];     /* samplePSA: woT; pPSA: wgT */
    double  u1, u2;
} ubdpt_bsdf_query;

typedef struct ubdpt_bsdf_result {
    double  f[3];      /* samplePSA value / scatter's bsdf_samplePSA */
    double  pdf;       /* pPSA value / scatter's pPSA */
    double  dir[3];    /* scatter's scat_dirT */
    int32_t ok;        /* scatter's return value */
    int32_t specular;
} ubdpt_bsdf_result;

/*
 * Connection probe: UBDPTRenderer::connectionFactor (uengine/ubdptrenderer.cpp:596-691), weight (:779-844, with
 * p_sp1 :693-735 and p_sm1 :737-777) and s0sample (:156-222) evaluated by the device code of the rendering kernels
 * on ONE pair of subpaths handed in as vertex arrays (UPathVertex, uengine/ubdptrenderer.h:9-24: world-space surface
 * point, throughput a, the two area pdfs, the specular flag; object = -1 for the lens vertex).
 * out[(s-1) * n_eye + (t-1)] for s = 1..n_light, t = 1..n_eye: connected = connectionFactor's return value (0 when an
 * end is specular, the ends do not see each other, or a cosine / BSDF test fails), c = the connection factor,
 * w = the MIS weight (only meaningful when connected). s0[3 * t .. 3 * t + 2] for t = 1..n_eye-1 = s0sample with
 * eye[t] as the vertex that hit an emitter and eye[0..t-1] as the subpath (zero when eye[t] is not on an emitter);
 * s0 may be NULL. At most 255 vertices per subpath.
 */
typedef struct ubdpt_path_vertex {
    double  pos[3], Ng[3], Ns[3], Ts[3], Bs[3];
    double  tex_u, tex_v;
    double  a[3];
    double  p_light_A, p_eye_A;
    int32_t object;     /* -1 for the lens vertex */
    int32_t lobe;       /* UBDPT_LOBE_* */
    int32_t specular;
    int32_t reserved;
} ubdpt_path_vertex;
typedef struct ubdpt_connection {
    int32_t connected;
    int32_t reserved;
    double  c[3];
    double  w;
} ubdpt_connection;
int ubdpt_connection_probe(ubdpt_context* ctx, const ubdpt_path_vertex* eye, uint32_t n_eye,
                           const ubdpt_path_vertex* light, uint32_t n_light, ubdpt_connection* out, double* s0);

int ubdpt_bsdf_eval_batch(ubdpt_context* ctx, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);
int ubdpt_bsdf_pdf_batch(ubdpt_context* ctx, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);
int ubdpt_bsdf_scatter_batch(ubdpt_context* ctx, const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);

#ifdef __cplusplus
}
#endif
#endif /* UBDPT_H */
