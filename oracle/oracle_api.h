/*
 * oracle_api.h — TEST INFRASTRUCTURE. C interface of the CPU oracle (bdpt_oracle.cpp).
 *
 * The oracle is a CPU restatement of the reference's BDPT hot path
 * (UEngine::renderPass -> UBDPTRenderer::renderPass -> renderPixel,
 *  uengine/uengine.cpp:144-180, uengine/ubdptrenderer.cpp:35-844) over the flat scene
 * description of include/ubdpt.h. It exists only to CHECK the HIP backend:
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, nothing else.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors, fixtures or scene files, and it
 * cannot be built in this environment (it needs GLM and assimp, neither vendored nor installed),
 * so no output of the reference itself pins this restatement. What pins it instead: closed-form
 * known-answer tests written from the reference's formulas, the aggregate figures of one probe run
 * of the reference's sources (mean radiance, rays and vertices per pixel-sample of a Cornell box),
 * and algorithmic invariants (tests/test_oracle_kat.py, test_oracle_anchors.py).
 */
#ifndef ORACLE_API_H
#define ORACLE_API_H

#include "../include/ubdpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

typedef struct oracle_counters {
    uint64_t pixel_samples;
    uint64_t closest_hit_rays;   /* UScene::intersectionPoint calls */
    uint64_t shadow_rays;        /* UScene::visibility calls */
    uint64_t eye_vertices;
    uint64_t light_vertices;
    uint64_t connections;        /* contributing (s,t) pairs */
    uint64_t splats;             /* t == 1 contributions */
    uint64_t capped_subpaths;
    uint64_t prims_tested;       /* sphere + triangle tests in closest-hit and shadow scans */
    uint64_t nodes_visited;      /* box tests of the per-mesh trees (Mode B only) */
    uint64_t prims_tested_closest; /* the part of prims_tested / nodes_visited spent on closest-hit rays */
    uint64_t nodes_visited_closest;
} oracle_counters;

/* One stored path vertex, world space (UPathVertex, uengine/ubdptrenderer.h:9-24): the layout of the backend's probe. */
typedef ubdpt_path_vertex oracle_vertex;

int  oracle_create(const ubdpt_scene_desc* scene, const ubdpt_render_params* params,
                   uint32_t max_path_vertices, oracle_scene** out);
void oracle_destroy(oracle_scene* s);

/* mode 0 (default): the reference's scans over every object and every face (uengine/uscene.cpp:41-66,
 * uapp/mesh.cpp:123-137). mode 1 ("Mode B"): the same object loop, but a mesh finds its closest / any blocking
 * face through a bounding-volume hierarchy built here; not in the reference, needed because the scan is not
 * runnable at 10^5..10^6 faces. Every hit is still decided by the reference's triangle test and tie rule, and the
 * tests assert that both modes report the same hits. */
int  oracle_set_accel(oracle_scene* s, int mode);

/* Adds passes first_pass + i*pass_stride (i < n_passes) into rgb (H x W x 3, row-major [y][x][rgb]). */
int  oracle_render_passes(oracle_scene* s, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride,
                          uint64_t seed, int num_threads, double* rgb, oracle_counters* counters);

/* renderPixel for one pixel of one pass, returning the two subpaths and the pixel's own measurement I
 * (splats to other pixels are returned as (x, y, r, g, b) rows, at most max_splats). */
int  oracle_trace_pixel(oracle_scene* s, uint64_t pass, uint64_t seed, uint32_t px, uint32_t py,
                        oracle_vertex* eye, uint32_t* n_eye, oracle_vertex* light, uint32_t* n_light,
                        double I[3], double* splats, uint32_t max_splats, uint32_t* n_splats);

/* connectionFactor / weight / s0sample (uengine/ubdptrenderer.cpp:596-844, :156-222) on one pair of subpaths given as
 * vertex arrays; same outputs as ubdpt_connection_probe (include/ubdpt.h). */
int  oracle_connection_probe(oracle_scene* s, const oracle_vertex* eye, uint32_t n_eye, const oracle_vertex* light,
                             uint32_t n_light, ubdpt_connection* out, double* s0);

int  oracle_intersect_batch(oracle_scene* s, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits);
int  oracle_visibility_batch(oracle_scene* s, const ubdpt_segment* segs, size_t n, uint8_t* visible);
int  oracle_get_object_info(oracle_scene* s, uint32_t object, double invW[16], double* area,
                            double* probability, double bounding_sphere[16]);
int  oracle_get_camera_info(oracle_scene* s, double view[16], double inv_view[16], double* image_plane_distance);

int  oracle_bsdf_eval_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);
int  oracle_bsdf_pdf_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);
int  oracle_bsdf_scatter_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r);

/* TextureImg::sample / TextureColor::sample (uapp/textureimg.cpp:56-82) on texture `tex` of the scene. */
int  oracle_texture_sample_batch(oracle_scene* s, uint32_t tex, const double* uv, size_t n, double* rgb);

/* The counter-based uniform generator shared (by specification) with the backend. */
double oracle_uniform(uint64_t seed, uint64_t pass, uint64_t pixel, uint32_t stream, uint32_t dim);

/* URng sampling maps (uengine/umath.cpp:22-86) for given uniforms.
 * which: 0 rect-stratified(num_strata, id, u1, u2) -> xy; 1 disk-stratified -> xy;
 *        2 hemisphere uniform -> xyz; 3 hemisphere cosine -> xyz; 4 sphere uniform -> xyz;
 *        5 triangle barycentric map (u1, u2) -> (1-s, s-m, m). */
int  oracle_sample_map(int which, uint64_t num_strata, uint64_t stratum, double u1, double u2, double out[3]);

/* UEngine::imageRGB (uengine/uengine.cpp:182-202): radiance = sum / curr_pass, CIE XYZ fit, sRGB with
 * gamma (uengine/uconverter.cpp). sum and rgb: npix x 3 doubles. UBDPT_NO_DATA when curr_pass == 0 or gamma <= 0. */
int  oracle_image_rgb(const double* sum, size_t npix, uint64_t curr_pass, double gamma, double* rgb);

#ifdef __cplusplus
}
#endif
#endif
