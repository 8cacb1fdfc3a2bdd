/*
 * bdpt_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), never linked into the shipped backend.
 *
 * Plain C++ restatement of the reference's bidirectional path tracer over the flat scene
 * description of include/ubdpt.h: brute-force intersection exactly in the reference's scan order,
 * double precision, every behavioural quirk of the reference kept (each one is marked "QUIRK").
 * Every function cites the reference lines it follows (paths relative to the reference root).
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors / fixtures / scene files and is
 * not buildable here (GLM and assimp missing), so this restatement is checked only against
 * closed-form cases and its own invariants (tests/test_oracle_*.py), not against reference output.
 *
 * Two deliberate departures from the reference, both statistically neutral:
 *  (1) Random numbers. The reference draws from one process-wide mt19937 seeded by
 *      std::random_device (uengine/umath.cpp:10-20), shared unsynchronised between threads, so no
 *      run is reproducible. Here every uniform is u(seed, pass, pixel, stream, dimension) from a
 *      counter-based hash, so oracle and backend trace the very same paths. The material lobe is
 *      drawn once for the closest hit instead of once per candidate object
 *      (uapp/object.cpp:22 inside the scan of uengine/uscene.cpp:48-60); i.i.d. draws make that
 *      equivalent in distribution.
 *  (2) Subpaths are capped at max_path_vertices (the reference is unbounded: Russian roulette only,
 *      uengine/ubdptrenderer.cpp:352-359).
 *
 * Build: g++ -O2 -ffp-contract=off (see oracle/Makefile).
 */
#include "oracle_api.h"
#include "omath.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

using namespace om;

namespace {

/* ------------------------------------------------------------------------------------------ */
/* counter-based uniform generator (specification shared with the backend)                     */
/* ------------------------------------------------------------------------------------------ */
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t path_key(uint64_t seed, uint64_t pass, uint64_t pixel) {
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (pass + 1));
    return mix64(k ^ (pixel * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}
inline double uniform01(uint64_t key, uint32_t stream, uint32_t dim) {
    uint64_t ctr = (((uint64_t)stream << 32) | dim) + 1;
    uint64_t h = mix64(key + 0x9E3779B97F4A7C15ULL * ctr);
    return (double)(h >> 11) * 0x1.0p-53; /* [0,1) like uniform_real_distribution<double>(0,1) */
}
enum { STREAM_EYE = 0, STREAM_LIGHT = 1 };
enum { DIM_LENS_X = 0, DIM_LENS_Y = 1, DIM_PIXEL_X = 2, DIM_PIXEL_Y = 3 };
enum { DIM_EMITTER = 0, DIM_POINT_A = 1, DIM_POINT_B = 2, DIM_POINT_C = 3, DIM_EMIT_ANGLE = 4, DIM_EMIT_V = 5 };
/* the h-th intersection of a subpath (h = 1, 2, ...) owns dimensions 8h .. 8h+3 */
enum { DIM_LOBE = 0, DIM_SCATTER_1 = 1, DIM_SCATTER_2 = 2, DIM_ROULETTE = 3 };
inline uint32_t hit_dim(uint32_t h, uint32_t which) { return 8u * h + which; }

struct Rng {
    uint64_t key;
    uint32_t stream;
    double operator()(uint32_t dim) const { return uniform01(key, stream, dim); }
};

/* ------------------------------------------------------------------------------------------ */
/* URng sampling maps — uengine/umath.cpp:22-86                                                */
/* ------------------------------------------------------------------------------------------ */
/* umath.cpp:22-34 */
dvec2 sampleUnitRectStratified(size_t num_strata, size_t stratum_id, double u1, double u2) {
    size_t num_divs = (size_t)std::sqrt((double)num_strata);
    double d = 1.0 / (double)num_divs;
    size_t x = stratum_id % num_divs;
    size_t y = stratum_id / num_divs;
    return {d * ((double)x + u1), d * ((double)y + u2)};
}
/* umath.cpp:36-44 */
dvec2 sampleUnitDiskStratified(size_t num_strata, size_t stratum_id, double u1, double u2) {
    dvec2 uv = sampleUnitRectStratified(num_strata, stratum_id, u1, u2);
    double theta = 2 * M_PI * uv.x;
    double r = std::sqrt(uv.y);
    return {r * std::cos(theta), r * std::sin(theta)};
}
/* umath.cpp:46-53 */
dvec3 samplePosHemUniform(double u1, double u2) {
    double angle = u1 * M_PI * 2.0;
    double v = u2;
    double s = std::sqrt(1 - v * v);
    return {std::cos(angle) * s, v, std::sin(angle) * s};
}
/* umath.cpp:55-63 */
dvec3 samplePosHemCos(double u1, double u2) {
    double angle = u1 * M_PI * 2.0;
    double s = u2;
    double y = std::sqrt(s);
    double r = std::sqrt(1 - s);
    return {r * std::cos(angle), y, r * std::sin(angle)};
}
/* umath.cpp:65-72 */
dvec3 sampleUnitSphereUniform(double u1, double u2) {
    double u = u1 * 2.0 * M_PI;
    double v = u2 * 2.0 - 1.0;
    double r = std::sqrt(1 - (v * v));
    return {std::cos(u) * r, v, std::sin(u) * r};
}
/* umath.cpp:74-86 */
dvec3 sampleTriangleUniform(dvec3 p0, dvec3 p1, dvec3 p2, double r1, double r2, double& u, double& v) {
    double s = std::sqrt(r1);
    double m = s * r2;
    u = s - m;
    v = m;
    return (1.0 - s) * p0 + (s - m) * p1 + m * p2;
}

/* ------------------------------------------------------------------------------------------ */
/* geometry — uengine/ugeometry.cpp                                                            */
/* ------------------------------------------------------------------------------------------ */
/* ugeometry.cpp:4-15 */
double triangleArea(dvec3 p0, dvec3 p1, dvec3 p2) {
    dvec3 AB = p1 - p0;
    dvec3 AC = p2 - p0;
    double l1 = length(AB);
    double l2 = length(AC);
    double c = dot(AB / l1, AC / l2);
    double s = std::sqrt(1 - c * c);
    return 0.5 * l1 * l2 * s;
}

struct Ray { /* URay, ugeometry.h:35-37 */
    dvec4 o, d;
    Ray() = default;
    Ray(dvec3 origin, dvec3 dir) : o(v4(origin, 1)), d(v4(dir, 0)) {}
    dvec3 origin() const { return v3(o); }
    dvec3 dir() const { return v3(d); }
    /* ugeometry.cpp:55-58: no renormalisation, so the local parameter equals the world parameter */
    Ray transform(const dmat4& T) const { return Ray(v3(T * o), v3(T * d)); }
};

/* ugeometry.cpp:70-94 */
bool intersectUnitSphere(const Ray& r, double& d) {
    double a = dot(r.dir(), r.dir());
    double b = 2 * dot(r.origin(), r.dir());
    double c = dot(r.origin(), r.origin()) - 1;
    double delta = (b * b) - (4 * a * c);
    if (delta < 0) return false;
    double sd = std::sqrt(delta);
    d = (-b - sd) / (2 * a);
    if (d > 0) return true;
    d = (-b + sd) / (2 * a);
    if (d < 0) return false;
    return true;
}
/* ugeometry.cpp:96-114: no parallel guard, no back-face cull, no tmax */
bool intersectTriangle(const Ray& r, dvec3 p0, dvec3 p1, dvec3 p2, double& d, double& u, double& v) {
    dvec3 e1 = p1 - p0;
    dvec3 e2 = p2 - p0;
    dvec3 m = r.origin() - p0;
    dvec3 c1 = cross(r.dir(), e2);
    dvec3 c2 = cross(m, e1);
    double a = dot(e1, c1);
    d = dot(e2, c2) / a;
    u = dot(m, c1) / a;
    v = dot(r.dir(), c2) / a;
    return d > 0 && u >= 0 && v >= 0 && u + v <= 1;
}
/* ugeometry.cpp:116-136 */
dvec3 transformPoint(const dmat4& T, dvec3 P) { return v3(T * v4(P, 1.0)); }
dvec3 transformVector(const dmat4& T, dvec3 V) { return normalize(v3(T * v4(V, 0.0))); }
dvec3 transformVectorT(const dmat4& T, dvec3 V) { return normalize(v3(v4(V, 0.0) * T)); }

/* ------------------------------------------------------------------------------------------ */
/* flattened scene                                                                             */
/* ------------------------------------------------------------------------------------------ */
struct MeshVertex { dvec3 pos; double tex_u; dvec3 normal; double tex_v; dvec3 tangent; }; /* uapp/mesh.h:13-20 */

struct Texture {
    uint32_t kind, width, height;
    dvec3 color;
    std::vector<dvec3> pixels; /* uapp/textureimg.cpp:24-33 */
};

struct Object {
    uint32_t kind, material;
    double mat_a, mat_b;
    uint32_t texture;
    bool is_emitter;
    int emitter_index;
    dmat4 W, invW;          /* uapp/object.cpp:5-12 */
    dvec3 power;            /* uapp/emitter.cpp:5 */
    double area;            /* uapp/emitter.cpp:6 */
    double probability;     /* uengine/uemitter.h:26-29 */
    size_t first_face, num_faces;
    dmat4 bounding_sphere;  /* uapp/mesh.cpp:85 */
};

struct SurfacePoint { /* USurfacePoint, uengine/uutils.h:22-35 (W/invW/bsdf via object + lobe) */
    int object = -1;
    int lobe = UBDPT_LOBE_NONE;
    long face = -1;
    dvec3 pos{0, 0, 0}, Ng{0, 0, 0}, Ns{0, 0, 0}, Ts{0, 0, 0}, Bs{0, 0, 0};
    double tex_u = 0, tex_v = 0;
};

struct PathVertex { /* UPathVertex, uengine/ubdptrenderer.h:9-24 */
    SurfacePoint sp;
    dvec3 a{0, 0, 0};
    bool specular = false;
    double p_light_A = 0, p_eye_A = 0;
};

struct Counters {
    uint64_t pixel_samples = 0, closest = 0, shadow = 0, eye_v = 0, light_v = 0, conn = 0, splats = 0, capped = 0,
             prims = 0, nodes = 0, prims_closest = 0, nodes_closest = 0;
};

/* ------------------------------------------------------------------------------------------ */
/* Mode B: a bounding-volume hierarchy per mesh. NOT in the reference, which tests every face  */
/* of a mesh for every ray (uapp/mesh.cpp:123-137, :170-180): at the sizes of the mesh          */
/* workloads that scan is not runnable, so the CPU baseline for them walks this tree instead.   */
/* The tree only decides WHICH faces are tested; every test is the reference's                  */
/* intersectTriangle on the same object-local ray, and the winner is the lexicographic minimum  */
/* of (d, face index), which is what the scan's strict '<' in face order selects. Boxes are     */
/* padded far beyond the rounding of the triangle test, so no face the scan accepts is skipped  */
/* (asserted against the scan on fixed ray batches by the tests).                               */
/* ------------------------------------------------------------------------------------------ */
struct BvhNode { /* 64 bytes: box, then either two children (count == 0: left, left + 1) or a run of faces */
    double lo[3], hi[3];
    uint32_t left_or_first, count;
    uint64_t pad;
};
struct MeshBvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> faces; /* face indices inside the mesh, leaf order */
};

} // namespace

struct oracle_scene {
    std::vector<Object> objects;
    std::vector<int> emitters; /* object indices, UScene::emitters() order */
    std::vector<MeshVertex> verts; /* 3 per face */
    std::vector<double> face_prob; /* uapp/mesh.cpp:88-106, indexed like faces */
    std::vector<Texture> textures;

    /* UBDPTRenderer members, uengine/ubdptrenderer.h:54-79 */
    size_t img_res_x, img_res_y, num_pixel_strata, num_lens_strata, min_depth;
    double focus_plane_distance, lens_radius;
    double image_plane_distance, image_plane_ratio, image_plane_area, pixel_width, pixel_height;
    double lens_area;
    dmat4 V, invV;
    const double m_W = 1.0;
    uint32_t max_path_vertices;
    int accel = 0;                /* 0: the reference's scans; 1: Mode B, a tree per mesh */
    std::vector<MeshBvh> bvh;     /* per object; empty for spheres and until Mode B is asked for */

    /* ---- Mode B build: binned surface-area split, leaves of at most 4 faces ---- */
    void buildMeshBvh(int oi) {
        const Object& o = objects[oi];
        MeshBvh& B = bvh[oi];
        const size_t n = o.num_faces;
        B.faces.resize(n);
        std::vector<double> flo(3 * n), fhi(3 * n), cen(3 * n);
        double mlo[3] = {1e300, 1e300, 1e300}, mhi[3] = {-1e300, -1e300, -1e300};
        for (size_t f = 0; f < n; f++) {
            B.faces[f] = (uint32_t)f;
            const MeshVertex* fv = &verts[3 * (o.first_face + f)];
            for (int k = 0; k < 3; k++) {
                double a = (&fv[0].pos.x)[k], b = (&fv[1].pos.x)[k], c = (&fv[2].pos.x)[k];
                flo[3 * f + k] = std::min(a, std::min(b, c));
                fhi[3 * f + k] = std::max(a, std::max(b, c));
                cen[3 * f + k] = 0.5 * (flo[3 * f + k] + fhi[3 * f + k]);
                mlo[k] = std::min(mlo[k], flo[3 * f + k]);
                mhi[k] = std::max(mhi[k], fhi[3 * f + k]);
            }
        }
        /* padding: 1e-7 of the mesh size, many orders above the rounding of intersectTriangle */
        const double pad = 1e-7 * std::max(mhi[0] - mlo[0], std::max(mhi[1] - mlo[1], mhi[2] - mlo[2])) + 1e-30;
        B.nodes.clear();
        B.nodes.reserve(n);
        B.nodes.push_back(BvhNode{});
        struct Job { uint32_t node, first, count; };
        std::vector<Job> jobs{{0, 0, (uint32_t)n}};
        constexpr int kBins = 16;
        while (!jobs.empty()) {
            Job j = jobs.back();
            jobs.pop_back();
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
            for (uint32_t i = j.first; i < j.first + j.count; i++) {
                uint32_t f = B.faces[i];
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], flo[3 * f + k]); hi[k] = std::max(hi[k], fhi[3 * f + k]);
                    clo[k] = std::min(clo[k], cen[3 * f + k]); chi[k] = std::max(chi[k], cen[3 * f + k]);
                }
            }
            BvhNode nd{};
            for (int k = 0; k < 3; k++) { nd.lo[k] = lo[k] - pad; nd.hi[k] = hi[k] + pad; }
            int axis = 0;
            if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
            if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
            const double ext = chi[axis] - clo[axis];
            if (j.count <= 4 || !(ext > 0)) { /* leaf (also: all centres coincide) */
                nd.left_or_first = j.first; nd.count = j.count;
                B.nodes[j.node] = nd;
                continue;
            }
            /* bins along the axis of the widest centroid extent; cost = area(left) * n(left) + area(right) * n(right) */
            uint32_t bc[kBins] = {0};
            double blo[kBins][3], bhi[kBins][3];
            for (int b = 0; b < kBins; b++) for (int k = 0; k < 3; k++) { blo[b][k] = 1e300; bhi[b][k] = -1e300; }
            const double scale = (double)kBins * (1.0 - 1e-9) / ext;
            for (uint32_t i = j.first; i < j.first + j.count; i++) {
                uint32_t f = B.faces[i];
                int b = (int)((cen[3 * f + axis] - clo[axis]) * scale);
                bc[b]++;
                for (int k = 0; k < 3; k++) { blo[b][k] = std::min(blo[b][k], flo[3 * f + k]); bhi[b][k] = std::max(bhi[b][k], fhi[3 * f + k]); }
            }
            auto area = [](const double* l, const double* h) {
                double x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
                return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + y * z + z * x);
            };
            double la[kBins], ra[kBins];
            uint32_t ln[kBins], rn[kBins];
            {
                double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
                uint32_t c = 0;
                for (int b = 0; b < kBins; b++) {
                    c += bc[b];
                    for (int k = 0; k < 3; k++) { l[k] = std::min(l[k], blo[b][k]); h[k] = std::max(h[k], bhi[b][k]); }
                    la[b] = area(l, h); ln[b] = c;
                }
                double l2[3] = {1e300, 1e300, 1e300}, h2[3] = {-1e300, -1e300, -1e300};
                c = 0;
                for (int b = kBins - 1; b >= 0; b--) {
                    c += bc[b];
                    for (int k = 0; k < 3; k++) { l2[k] = std::min(l2[k], blo[b][k]); h2[k] = std::max(h2[k], bhi[b][k]); }
                    ra[b] = area(l2, h2); rn[b] = c;
                }
            }
            int best = -1;
            double best_cost = 1e300;
            for (int b = 0; b + 1 < kBins; b++) {
                if (ln[b] == 0 || rn[b + 1] == 0) continue;
                double cost = la[b] * ln[b] + ra[b + 1] * rn[b + 1];
                if (cost < best_cost) { best_cost = cost; best = b; }
            }
            uint32_t mid;
            if (best < 0) mid = j.first + j.count / 2; /* everything in one bin: split the run in the middle */
            else {
                uint32_t* beg = B.faces.data() + j.first;
                uint32_t* m = std::partition(beg, beg + j.count, [&](uint32_t f) {
                    return (int)((cen[3 * f + axis] - clo[axis]) * scale) <= best;
                });
                mid = (uint32_t)(m - B.faces.data());
            }
            nd.left_or_first = (uint32_t)B.nodes.size(); nd.count = 0;
            B.nodes[j.node] = nd;
            B.nodes.push_back(BvhNode{});
            B.nodes.push_back(BvhNode{});
            jobs.push_back({nd.left_or_first, j.first, mid - j.first});
            jobs.push_back({nd.left_or_first + 1, mid, j.first + j.count - mid});
        }
    }
    void setAccel(int mode) {
        accel = mode;
        if (mode == 1 && bvh.empty()) {
            bvh.resize(objects.size());
            for (int o = 0; o < (int)objects.size(); o++)
                if (objects[o].kind == UBDPT_MODEL_MESH) buildMeshBvh(o);
        }
    }
    /* entry and exit parameter of the ray through a padded box; the interval is widened by 1e-12 relative */
    static bool slab(const BvhNode& n, const double o[3], const double inv[3], double tmax, double& tnear) {
        double t0 = -std::numeric_limits<double>::infinity(), t1 = std::numeric_limits<double>::infinity();
        for (int k = 0; k < 3; k++) {
            double a = (n.lo[k] - o[k]) * inv[k], b = (n.hi[k] - o[k]) * inv[k];
            if (a != a || b != b) continue; /* 0 * inf: the origin lies on a plane of a slab the ray is parallel to */
            if (a > b) { double t = a; a = b; b = t; }
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
        }
        t0 -= std::fabs(t0) * 1e-12;
        t1 += std::fabs(t1) * 1e-12;
        tnear = t0;
        return t0 <= t1 && t1 >= 0 && t0 <= tmax;
    }
    /* Closest face of mesh `o` for the local ray: lexicographic minimum of (d, face). Returns false when no face is hit. */
    bool bvhClosestFace(int oi, const Ray& rayL, double& ci_d, double& ci_u, double& ci_v, size_t& ci_face, Counters* cn) const {
        const Object& o = objects[oi];
        const MeshBvh& B = bvh[oi];
        const dvec3 ro = rayL.origin(), rd = rayL.dir();
        const double org[3] = {ro.x, ro.y, ro.z};
        const double inv[3] = {1.0 / rd.x, 1.0 / rd.y, 1.0 / rd.z};
        const double inf = std::numeric_limits<double>::infinity();
        ci_d = inf; ci_u = ci_v = 0; ci_face = 0;
        bool hit = false;
        uint32_t stack[128];
        int sp = 0;
        double tn;
        if (cn) cn->nodes++;
        if (!slab(B.nodes[0], org, inv, inf, tn)) return false;
        uint32_t node = 0;
        for (;;) {
            const BvhNode& n = B.nodes[node];
            if (n.count) {
                for (uint32_t i = n.left_or_first; i < n.left_or_first + n.count; i++) {
                    const uint32_t f = B.faces[i];
                    const MeshVertex* face = &verts[3 * (o.first_face + f)];
                    double d, u, v;
                    if (cn) cn->prims++;
                    if (intersectTriangle(rayL, face[0].pos, face[1].pos, face[2].pos, d, u, v))
                        if (d < ci_d || (d == ci_d && f < ci_face)) { ci_d = d; ci_u = u; ci_v = v; ci_face = f; hit = true; }
                }
            } else {
                double ta, tb;
                const uint32_t a = n.left_or_first, b = a + 1;
                if (cn) cn->nodes += 2;
                const bool ha = slab(B.nodes[a], org, inv, ci_d, ta), hb = slab(B.nodes[b], org, inv, ci_d, tb);
                if (ha && hb) {
                    if (sp >= 126) return hit; /* cannot happen: depth is bounded by the face count's logarithm times a constant */
                    if (ta <= tb) { stack[sp++] = b; node = a; } else { stack[sp++] = a; node = b; }
                    continue;
                }
                if (ha) { node = a; continue; }
                if (hb) { node = b; continue; }
            }
            /* pop; a box entered behind the best hit so far cannot hold a smaller (d, face) */
            for (;;) {
                if (sp == 0) return hit;
                node = stack[--sp];
                if (cn) cn->nodes++;
                if (slab(B.nodes[node], org, inv, ci_d, tn)) break;
            }
        }
    }
    /* Does any face of mesh `o` block the local ray before `dist`? (mesh.cpp:170-187 finds the smallest d of all
     * accepted faces; it is below dist exactly when some accepted face's d is, and accepted faces have d > 0.) */
    bool bvhMinBelow(int oi, const Ray& rayL, double dist, double& min_d, Counters* cn) const {
        const Object& o = objects[oi];
        const MeshBvh& B = bvh[oi];
        const dvec3 ro = rayL.origin(), rd = rayL.dir();
        const double org[3] = {ro.x, ro.y, ro.z};
        const double inv[3] = {1.0 / rd.x, 1.0 / rd.y, 1.0 / rd.z};
        uint32_t stack[128];
        int sp = 0;
        double tn;
        if (cn) cn->nodes++;
        if (!slab(B.nodes[0], org, inv, dist, tn)) return false;
        stack[sp++] = 0;
        while (sp) {
            const BvhNode& n = B.nodes[stack[--sp]];
            if (n.count) {
                for (uint32_t i = n.left_or_first; i < n.left_or_first + n.count; i++) {
                    const MeshVertex* face = &verts[3 * (o.first_face + B.faces[i])];
                    double d, u, v;
                    if (cn) cn->prims++;
                    if (intersectTriangle(rayL, face[0].pos, face[1].pos, face[2].pos, d, u, v) && d < dist) { min_d = d; return true; }
                }
            } else {
                const uint32_t a = n.left_or_first;
                if (cn) cn->nodes += 2;
                if (sp < 126 && slab(B.nodes[a], org, inv, dist, tn)) stack[sp++] = a;
                if (sp < 126 && slab(B.nodes[a + 1], org, inv, dist, tn)) stack[sp++] = a + 1;
            }
        }
        return false;
    }

    /* --- uapp/mesh.cpp:38-86 --- */
    dmat4 computeBoundingSphere(size_t first, size_t n) const {
        double max_d = 0;
        dvec3 x = verts[3 * first].pos;
        dvec3 y{0, 0, 0}, z{0, 0, 0};
        for (size_t f = first; f < first + n; f++)
            for (size_t v = 0; v < 3; v++) {
                double d = distance(verts[3 * f + v].pos, x);
                if (d > max_d) { max_d = d; y = verts[3 * f + v].pos; }
            }
        max_d = 0;
        for (size_t f = first; f < first + n; f++)
            for (size_t v = 0; v < 3; v++) {
                double d = distance(verts[3 * f + v].pos, y);
                if (d > max_d) { max_d = d; z = verts[3 * f + v].pos; }
            }
        dvec3 center = 0.5 * (y + z);
        double radius = 0.5 * max_d;
        for (size_t f = first; f < first + n; f++)
            for (size_t v = 0; v < 3; v++) {
                double d = distance(verts[3 * f + v].pos, center);
                if (d > radius) radius = d;
            }
        return inverse(scale(translate(identity4(), center), dvec3{radius, radius, radius}));
    }

    /* --- uapp/textureimg.cpp:56-82 / uapp/texturecolor.cpp:8-11 --- */
    dvec3 textureSample(uint32_t t, double u, double v) const {
        const Texture& tx = textures[t];
        if (tx.kind == UBDPT_TEX_COLOR) return tx.color;
        /* QUIRK: v is "wrapped" with size_t(u), not size_t(v) (textureimg.cpp:60) */
        u = u - (double)(size_t)u;
        v = v - (double)(size_t)u;
        double x = u * (double)(tx.width - 1);
        double y = v * (double)(tx.height - 1);
        size_t x1 = (size_t)std::floor(x), x2 = (size_t)std::ceil(x);
        size_t y1 = (size_t)std::floor(y), y2 = (size_t)std::ceil(y);
        double tx_ = x - (double)x1;
        double ty_ = y - (double)y1;
        /* the reference asserts these indices are in range (textureimg.cpp:73-76); out-of-range uv is
         * undefined there, here the indices are clamped so the oracle never reads out of bounds */
        size_t wmax = tx.width - 1, hmax = tx.height - 1;
        if (x1 > wmax) x1 = wmax; if (x2 > wmax) x2 = wmax;
        if (y1 > hmax) y1 = hmax; if (y2 > hmax) y2 = hmax;
        const std::vector<dvec3>& p = tx.pixels;
        size_t w = tx.width;
        /* QUIRK: weights are swapped — tx multiplies the x1 texel (textureimg.cpp:78-81) */
        dvec3 c1 = tx_ * p[w * y1 + x1] + (1.0 - tx_) * p[w * y1 + x2];
        dvec3 c2 = tx_ * p[w * y2 + x1] + (1.0 - tx_) * p[w * y2 + x2];
        return ty_ * c1 + (double)(1.0f - ty_) * c2;
    }

    /* --- uapp/implicitsphere.cpp:3-18 --- */
    static bool sphereLocalIntersection(const Ray& rayL, SurfacePoint& sp, double& d) {
        if (!intersectUnitSphere(rayL, d)) return false;
        sp.tex_u = 0;
        sp.tex_v = 0;
        sp.pos = rayL.origin() + d * rayL.dir();
        sp.Ns = sp.Ng = normalize(sp.pos);
        sp.Ts = normalize(-sp.Ns + dvec3{0, 0, 1.0 / sp.Ns.z}); /* QUIRK: NaN on the equator Ns.z == 0 */
        sp.Bs = normalize(cross(sp.Ns, sp.Ts));
        sp.face = -1;
        return true;
    }
    /* --- uapp/mesh.cpp:108-159 --- */
    bool meshLocalIntersection(const Object& o, const Ray& rayL, SurfacePoint& sp, double& sp_d, Counters* cn) const {
        double ci_d = std::numeric_limits<double>::infinity(), ci_u = 0, ci_v = 0;
        size_t ci_face = 0;
        double d, u, v;
        bool hit = false;
        if (!intersectUnitSphere(rayL.transform(o.bounding_sphere), d)) return false;
        if (accel == 1) hit = bvhClosestFace((int)(&o - objects.data()), rayL, ci_d, ci_u, ci_v, ci_face, cn); /* Mode B */
        else
        for (size_t f = 0; f < o.num_faces; f++) {
            const MeshVertex* face = &verts[3 * (o.first_face + f)];
            if (cn) cn->prims++;
            if (intersectTriangle(rayL, face[0].pos, face[1].pos, face[2].pos, d, u, v)) {
                if (d < ci_d) { ci_d = d; ci_u = u; ci_v = v; ci_face = f; hit = true; }
            }
        }
        if (!hit) return false;
        sp_d = ci_d;
        const MeshVertex* f = &verts[3 * (o.first_face + ci_face)];
        sp.tex_u = (1.0 - ci_u - ci_v) * f[0].tex_u + ci_u * f[1].tex_u + ci_v * f[2].tex_u;
        sp.tex_v = (1.0 - ci_u - ci_v) * f[0].tex_v + ci_u * f[1].tex_v + ci_v * f[2].tex_v;
        sp.pos = rayL.origin() + ci_d * rayL.dir();
        sp.Ns = normalize((1.0 - ci_u - ci_v) * f[0].normal + ci_u * f[1].normal + ci_v * f[2].normal);
        sp.Ng = normalize(cross(f[1].pos - f[0].pos, f[2].pos - f[0].pos));
        if (dot(sp.Ns, sp.Ng) < 0) sp.Ng *= -1.0;
        sp.Ts = normalize((1 - ci_u - ci_v) * f[0].tangent + ci_u * f[1].tangent + ci_v * f[2].tangent);
        sp.Bs = normalize(cross(sp.Ns, sp.Ts));
        sp.face = (long)ci_face;
        return true;
    }
    /* --- uapp/mesh.cpp:161-189 --- */
    bool meshIntersects(const Object& o, const Ray& rayL, double& d, Counters* cn) const {
        double u, v;
        double min_d = std::numeric_limits<double>::infinity();
        bool hit = false;
        if (!intersectUnitSphere(rayL.transform(o.bounding_sphere), d)) return false;
        for (size_t f = 0; f < o.num_faces; f++) {
            const MeshVertex* face = &verts[3 * (o.first_face + f)];
            if (cn) cn->prims++;
            if (intersectTriangle(rayL, face[0].pos, face[1].pos, face[2].pos, d, u, v)) {
                if (d < min_d) { min_d = d; hit = true; }
            }
        }
        if (hit) { d = min_d; return true; }
        return false;
    }
    /* Mode B form of Object::intersects for UScene::visibility: does the object block the ray before `dist`? */
    bool objectBlocks(int oi, const Ray& ray, double dist, Counters* cn) const {
        const Object& o = objects[oi];
        double d;
        if (o.kind == UBDPT_MODEL_SPHERE || accel != 1) return objectIntersects(oi, ray, d, cn) && (d < dist) && (d > 0);
        Ray rayL = ray.transform(o.invW);
        if (!intersectUnitSphere(rayL.transform(o.bounding_sphere), d)) return false;
        return bvhMinBelow(oi, rayL, dist, d, cn);
    }
    /* --- uapp/object.cpp:14-25 (the lobe is drawn later, for the closest hit only) --- */
    bool objectIntersectionPoint(int oi, const Ray& ray, SurfacePoint& sp, double& d, Counters* cn) const {
        const Object& o = objects[oi];
        Ray rayL = ray.transform(o.invW);
        if (o.kind == UBDPT_MODEL_SPHERE) {
            if (cn) cn->prims++;
            return sphereLocalIntersection(rayL, sp, d);
        }
        return meshLocalIntersection(o, rayL, sp, d, cn);
    }
    /* --- uapp/object.cpp:27-30 --- */
    bool objectIntersects(int oi, const Ray& ray, double& d, Counters* cn) const {
        const Object& o = objects[oi];
        Ray rayL = ray.transform(o.invW);
        if (o.kind == UBDPT_MODEL_SPHERE) {
            if (cn) cn->prims++;
            return intersectUnitSphere(rayL, d);
        }
        return meshIntersects(o, rayL, d, cn);
    }
    /* --- uengine/uscene.cpp:41-66: strict '<' so the first object in the list wins ties --- */
    bool intersectionPoint(const Ray& ray, SurfacePoint& closest_sp, double* out_d, Counters* cn) const {
        SurfacePoint sp;
        double min_d = std::numeric_limits<double>::infinity();
        bool hit = false;
        double d;
        if (cn) cn->closest++;
        const uint64_t prims0 = cn ? cn->prims : 0, nodes0 = cn ? cn->nodes : 0;
        for (int o = 0; o < (int)objects.size(); o++) {
            if (objectIntersectionPoint(o, ray, sp, d, cn)) {
                if (d < min_d) {
                    min_d = d;
                    closest_sp = sp;
                    closest_sp.object = o;
                    hit = true;
                }
            }
        }
        if (cn) { cn->prims_closest += cn->prims - prims0; cn->nodes_closest += cn->nodes - nodes0; }
        if (out_d) *out_d = min_d;
        return hit;
    }
    /* --- uengine/uscene.cpp:21-39 --- */
    bool visibility(dvec3 p0, dvec3 p1, Counters* cn) const {
        double d;
        double points_distance = length(p1 - p0);
        Ray ray(p0, (p1 - p0) / points_distance);
        if (cn) cn->shadow++;
        if (accel == 1) {
            for (int o = 0; o < (int)objects.size(); o++)
                if (objectBlocks(o, ray, points_distance, cn)) return false;
            return true;
        }
        for (int o = 0; o < (int)objects.size(); o++) {
            if (objectIntersects(o, ray, d, cn)) {
                if ((d < points_distance) && (d > 0)) return false;
            }
        }
        return true;
    }

    /* --- uapp/material.h:25-31,45-48,66-79,97-100: lobe choice with the uniform r --- */
    int pickLobe(int oi, double r) const {
        const Object& o = objects[oi];
        switch (o.material) {
        case UBDPT_MAT_LATEX_PAINT: /* QUIRK: 20 % absorbed and not compensated */
            return (r < 0.8) ? UBDPT_LOBE_LAMBERTIAN : UBDPT_LOBE_NONE;
        case UBDPT_MAT_GLOSSY:
            if (r < o.mat_a) return UBDPT_LOBE_LAMBERTIAN;
            r -= o.mat_a;
            if (r < o.mat_b) return UBDPT_LOBE_MIRROR;
            return UBDPT_LOBE_NONE;
        case UBDPT_MAT_PERFECT_MIRROR: return UBDPT_LOBE_MIRROR;
        case UBDPT_MAT_DIELECTRIC: return UBDPT_LOBE_DIELECTRIC;
        }
        return UBDPT_LOBE_NONE;
    }
};

namespace {

/* ------------------------------------------------------------------------------------------ */
/* BSDFs — uengine/ubsdf*.cpp. All directions in tangent space (y = shading normal), PSA measure */
/* ------------------------------------------------------------------------------------------ */
struct BsdfInfo { dvec3 Ng, Ns, Ts, Bs; double eta_t; }; /* UBsdfSurfaceInfo, ubsdf.h:8-20; eta_t = 1 (ubsdf.cpp:13) */

inline dmat3 TNBof(const BsdfInfo& i) { return {{i.Ts, i.Ns, i.Bs}}; }

/* ubsdfdielectric.cpp:3-9 */
double fersnel(double eta_i, double eta_t, double cos_i, double cos_t) {
    double rp = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    double rs = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    return 0.5 * (rp * rp + rs * rs);
}

/* UBsdf::samplePSA */
dvec3 bsdfSamplePSA(int lobe, double eta, dvec3 tex, const BsdfInfo& info, dvec3 wiT, dvec3 woT) {
    dmat3 TNB = TNBof(info);
    if (lobe == UBDPT_LOBE_LAMBERTIAN || lobe == UBDPT_LOBE_MIRROR) {
        /* ubsdflambertian.cpp:10-26, ubsdfperfectmirror.cpp:8-24 */
        dvec3 wiL = normalize(TNB * wiT);
        dvec3 woL = normalize(TNB * woT);
        if (dot(info.Ng, wiL) * dot(info.Ng, woL) <= 0) return {0, 0, 0};
        if (wiT.y * woT.y <= 0) return {0, 0, 0};
        if (lobe == UBDPT_LOBE_LAMBERTIAN) return (1.0 / M_PI) * tex;
        return tex;
    }
    if (lobe == UBDPT_LOBE_DIELECTRIC) {
        /* ubsdfdielectric.cpp:17-62 */
        double n, nt, Ny;
        if (wiT.y < 0) { Ny = -1.0; n = eta; nt = info.eta_t; }
        else           { Ny = 1.0;  n = info.eta_t; nt = eta; }
        double c = ((n * n) / (nt * nt)) * (1.0 - wiT.y * wiT.y);
        if (c > 1.0) {
            if (wiT.y * woT.y <= 0) return {0, 0, 0};
            return tex;
        }
        double R = fersnel(n, nt, Ny * wiT.y, std::sqrt(1.0 - c));
        double T = 1.0 - R;
        if (wiT.y * woT.y <= 0) return T * tex;
        return R * tex;
    }
    return {0, 0, 0};
}

/* UBsdf::pPSA */
double bsdfPPSA(int lobe, double eta, const BsdfInfo& info, dvec3 wsT, dvec3 wgT) {
    dmat3 TNB = TNBof(info);
    if (lobe == UBDPT_LOBE_LAMBERTIAN || lobe == UBDPT_LOBE_MIRROR) {
        /* ubsdflambertian.cpp:28-49 (cosine weighted, uapp/material.h:22,60), ubsdfperfectmirror.cpp:26-42 */
        dvec3 wsL = normalize(TNB * wsT);
        dvec3 wgL = normalize(TNB * wgT);
        if (dot(info.Ng, wsL) * dot(info.Ng, wgL) <= 0) return 0;
        if (wsT.y * wgT.y <= 0) return 0;
        return lobe == UBDPT_LOBE_LAMBERTIAN ? (1.0 / M_PI) : 1.0;
    }
    if (lobe == UBDPT_LOBE_DIELECTRIC) {
        /* ubsdfdielectric.cpp:64-108 */
        double n, nt, Ny;
        if (wgT.y < 0) { Ny = -1.0; n = eta; nt = info.eta_t; }
        else           { Ny = 1.0;  n = info.eta_t; nt = eta; }
        double c = ((n * n) / (nt * nt)) * (1.0 - wgT.y * wgT.y);
        if (c > 1.0) return (wsT.y * wgT.y <= 0) ? 0.0 : 1.0;
        double R = fersnel(n, nt, Ny * wgT.y, std::sqrt(1.0 - c));
        double T = 1.0 - R;
        return (wsT.y * wgT.y <= 0) ? T : R;
    }
    return 0;
}

/* UBsdf::scatter; u1,u2 are the uniforms the reference would draw, in its order */
bool bsdfScatter(int lobe, double eta, dvec3 tex, const BsdfInfo& info, dvec3 w, double u1, double u2,
                 dvec3& scat_dirT, double& pPSA, dvec3& fs, bool& specular) {
    dmat3 TNB = TNBof(info);
    if (lobe == UBDPT_LOBE_LAMBERTIAN) {
        /* ubsdflambertian.cpp:51-81, cosine weighted */
        if (dot(w, info.Ns) * dot(w, info.Ng) <= 0) return false;
        dvec3 wT = normalize(w * TNB);
        scat_dirT = samplePosHemCos(u1, u2);
        pPSA = 1.0 / M_PI;
        if (wT.y < 0) scat_dirT *= -1.0;
        specular = false;
        fs = (1.0 / M_PI) * tex;
        return true;
    }
    if (lobe == UBDPT_LOBE_MIRROR) {
        /* ubsdfperfectmirror.cpp:44-66 */
        if (dot(w, info.Ns) * dot(w, info.Ng) <= 0) return false;
        dvec3 wT = normalize(w * TNB);
        scat_dirT = normalize(reflect(-wT, dvec3{0, 1, 0}));
        if (wT.y < 0) scat_dirT *= -1.0; /* QUIRK: reflected direction negated below the surface */
        specular = true;
        pPSA = 1.0;
        fs = tex;
        return true;
    }
    if (lobe == UBDPT_LOBE_DIELECTRIC) {
        /* ubsdfdielectric.cpp:110-174 */
        if (dot(w, info.Ns) * dot(w, info.Ng) <= 0) return false;
        specular = true;
        dvec3 wT = normalize(w * TNB);
        double n, nt;
        dvec3 N;
        if (wT.y < 0) { N = {0.0, -1.0, 0.0}; n = eta; nt = info.eta_t; }
        else          { N = {0.0, 1.0, 0.0};  n = info.eta_t; nt = eta; }
        double eta_r = n / nt;
        double c = (eta_r * eta_r) * (1.0 - wT.y * wT.y);
        double R, T, c1 = 0;
        if (c > 1.0) R = 1.0;
        else { c1 = std::sqrt(1.0 - c); R = fersnel(n, nt, N.y * wT.y, c1); }
        T = 1.0 - R;
        if (u1 < R) {
            scat_dirT = normalize(reflect(-wT, N));
            pPSA = R;
            fs = R * tex;
        } else {
            scat_dirT = normalize(eta_r * (-wT) - N * (eta_r * dot(N, -wT) + c1));
            pPSA = T;
            fs = T * tex;
        }
        return true;
    }
    return false;
}

/* ------------------------------------------------------------------------------------------ */
/* the integrator — uengine/ubdptrenderer.cpp                                                  */
/* ------------------------------------------------------------------------------------------ */
struct Tracer {
    const oracle_scene& S;
    Counters* cn;
    explicit Tracer(const oracle_scene& s, Counters* c) : S(s), cn(c) {}

    BsdfInfo infoOf(const SurfacePoint& sp) const { return {sp.Ng, sp.Ns, sp.Ts, sp.Bs, 1.0}; } /* ubsdf.cpp:5-18 */
    double etaOf(const SurfacePoint& sp) const { return S.objects[sp.object].mat_a; }
    dvec3 texOf(const SurfacePoint& sp) const { return S.textureSample(S.objects[sp.object].texture, sp.tex_u, sp.tex_v); }

    /* ubdptrenderer.cpp:156-222 */
    dvec3 s0sample(const std::vector<PathVertex>& subpath, const PathVertex& emitter_vertex) const {
        if (emitter_vertex.sp.object < 0) return {0, 0, 0};
        const Object& e = S.objects[emitter_vertex.sp.object];
        if (!e.is_emitter) return {0, 0, 0};
        const PathVertex& curr_vertex = subpath.back();
        double w = 1.0;
        double ratio = (e.probability / e.area) / emitter_vertex.p_eye_A;
        if (!curr_vertex.specular) w += ratio * ratio;

        dvec3 edge = curr_vertex.sp.pos - emitter_vertex.sp.pos;
        double edge_length2 = dot(edge, edge);
        edge /= std::sqrt(edge_length2);
        double d1 = dot(emitter_vertex.sp.Ns, edge);
        double d2 = dot(curr_vertex.sp.Ns, -edge);
        double G = std::abs(d1 * d2) / edge_length2;
        double p_light_psa = 1.0 / (2.0 * M_PI * std::abs(d1));
        ratio *= (p_light_psa * G) / curr_vertex.p_eye_A;
        if (!curr_vertex.specular) {
            if (subpath.size() > 1) {
                if (!subpath[subpath.size() - 2].specular) w += ratio * ratio;
            } else
                w += ratio * ratio;
        }
        if (subpath.size() > 2)
            for (int v = (int)subpath.size() - 2; v > 0; v--) {
                ratio *= subpath[v].p_light_A / subpath[v].p_eye_A;
                if (subpath[v].specular) continue;
                if (v > 0)
                    if (subpath[v - 1].specular) continue;
                w += ratio * ratio;
            }
        w = 1.0 / w;
        dvec3 c = (e.power / e.area) * p_light_psa;
        return w * c * emitter_vertex.a;
    }

    /* steps shared by both subpath generators, ubdptrenderer.cpp:303-320 / 493-510:
     * flip the frame for transmission, offset along Ng in LOCAL space, move everything to world */
    void finishVertex(PathVertex& nv, dvec3 next_dirT) const {
        const Object& o = S.objects[nv.sp.object];
        if (next_dirT.y < 0) {
            nv.sp.Ng *= -1.0; nv.sp.Ns *= -1.0; nv.sp.Ts *= -1.0; nv.sp.Bs *= -1.0;
        }
        nv.sp.pos += 0.00001 * nv.sp.Ng; /* QUIRK: local-space offset, scales with the object */
        nv.sp.pos = transformPoint(o.W, nv.sp.pos);
        nv.sp.Ng = transformVectorT(o.invW, nv.sp.Ng);
        nv.sp.Ns = transformVectorT(o.invW, nv.sp.Ns);
        nv.sp.Ts = transformVectorT(o.invW, nv.sp.Ts);
        nv.sp.Bs = transformVectorT(o.invW, nv.sp.Bs);
    }

    /* ubdptrenderer.cpp:224-407 */
    dvec3 computeEyeSubpath(std::vector<PathVertex>& subpath, size_t px, size_t py, size_t lens_sample_id,
                            size_t pixel_sample_id, const Rng& rng) const {
        subpath.clear();
        dvec3 I{0, 0, 0};
        dvec2 disk = sampleUnitDiskStratified(S.num_lens_strata, lens_sample_id, rng(DIM_LENS_X), rng(DIM_LENS_Y));
        dvec3 lens_pointV{S.lens_radius * disk.x, S.lens_radius * disk.y, 0};

        PathVertex lens_vertex{};
        lens_vertex.a = {S.m_W, S.m_W, S.m_W};
        lens_vertex.sp.pos = v3(S.invV * v4(lens_pointV, 1.0));
        lens_vertex.sp.Ns = lens_vertex.sp.Ng = transformVector(S.invV, {0, 0, 1});
        lens_vertex.sp.Ts = transformVector(S.invV, {1, 0, 0});
        lens_vertex.sp.Bs = transformVector(S.invV, {0, 1, 0});
        lens_vertex.p_eye_A = 1.0 / S.lens_area; /* QUIRK: lens_area is never assigned -> 0 -> +inf */
        lens_vertex.specular = false;
        subpath.push_back(lens_vertex);

        dvec2 pixel_point =
            sampleUnitRectStratified(S.num_pixel_strata, pixel_sample_id, rng(DIM_PIXEL_X), rng(DIM_PIXEL_Y));
        dvec3 image_pointV{-S.image_plane_ratio + ((double)px + pixel_point.x) * S.pixel_width,
                           1.0 - ((double)py + pixel_point.y) * S.pixel_height, S.image_plane_distance};
        image_pointV = normalize(image_pointV);
        dvec3 focus_plane_pointV = image_pointV * (S.focus_plane_distance / image_pointV.z);
        dvec3 eye_ray_dirW = transformVector(S.invV, focus_plane_pointV - lens_pointV);
        Ray ray(lens_vertex.sp.pos, eye_ray_dirW);

        uint32_t h = 1; /* index of the intersection being processed */
        PathVertex next_vertex{};
        if (!S.intersectionPoint(ray, next_vertex.sp, nullptr, cn)) return {0, 0, 0};
        next_vertex.sp.lobe = S.pickLobe(next_vertex.sp.object, rng(hit_dim(h, DIM_LOBE)));
        next_vertex.a = lens_vertex.a;
        next_vertex.p_eye_A = 1.0 / S.image_plane_area;

        while (true) {
            const Object& obj = S.objects[next_vertex.sp.object];
            if (next_vertex.sp.lobe == UBDPT_LOBE_NONE) {
                /* :273-283 */
                next_vertex.sp.pos = transformPoint(obj.W, next_vertex.sp.pos);
                next_vertex.sp.Ns = transformVectorT(obj.invW, next_vertex.sp.Ns);
                I += s0sample(subpath, next_vertex);
                break;
            }
            if (subpath.size() >= S.max_path_vertices) { if (cn) cn->capped++; break; } /* cap (not in reference) */

            /* QUIRK: float TNB (:287-290) */
            fmat3 TNB{{to_f(next_vertex.sp.Ts), to_f(next_vertex.sp.Ns), to_f(next_vertex.sp.Bs)}};
            BsdfInfo scatter_info = infoOf(next_vertex.sp);
            dvec3 w = transformVector(obj.invW, -ray.dir());
            double p_psa;
            dvec3 fs, next_dirT;
            dvec3 tex = texOf(next_vertex.sp);
            if (!bsdfScatter(next_vertex.sp.lobe, etaOf(next_vertex.sp), tex, scatter_info, w,
                             rng(hit_dim(h, DIM_SCATTER_1)), rng(hit_dim(h, DIM_SCATTER_2)), next_dirT, p_psa, fs,
                             next_vertex.specular))
                break;
            finishVertex(next_vertex, next_dirT);
            I += s0sample(subpath, next_vertex);
            subpath.push_back(next_vertex);

            /* :332-336 */
            ray = Ray(subpath.back().sp.pos, transformVector(obj.W, to_d(TNB * to_f(next_dirT))));
            uint32_t h_curr = h;
            h++;
            if (!S.intersectionPoint(ray, next_vertex.sp, nullptr, cn)) break;
            next_vertex.sp.lobe = S.pickLobe(next_vertex.sp.object, rng(hit_dim(h, DIM_LOBE)));

            PathVertex& curr_vertex = subpath.back();
            PathVertex& prev_vertex = subpath[subpath.size() - 2];
            double fs_sum = fs.x + fs.y + fs.z;
            if (!(p_psa > 0) || !(fs_sum > 0)) break;
            double q;
            if (subpath.size() < S.min_depth) q = 1.0;
            else {
                q = std::min(1.0, (fs_sum / 3.0) / p_psa);
                if (rng(hit_dim(h_curr, DIM_ROULETTE)) > q) break;
            }
            if (curr_vertex.specular) {
                next_vertex.a = curr_vertex.a * fs;
                next_vertex.p_eye_A = q * p_psa;
                prev_vertex.p_light_A = q * p_psa;
            } else {
                dvec3 a = fs / p_psa;
                double G_prev, G_next, d1, d2, edge_length2;
                dvec3 edge;
                edge = prev_vertex.sp.pos - curr_vertex.sp.pos;
                edge_length2 = dot(edge, edge);
                edge /= std::sqrt(edge_length2);
                d1 = dot(edge, curr_vertex.sp.Ns);
                d2 = dot(-edge, prev_vertex.sp.Ns);
                G_prev = std::abs(d1 * d2) / edge_length2;
                /* QUIRK: next_vertex.sp is still in the hit object's LOCAL space here (:389-393) */
                edge = next_vertex.sp.pos - curr_vertex.sp.pos;
                edge_length2 = dot(edge, edge);
                edge /= std::sqrt(edge_length2);
                d1 = dot(edge, curr_vertex.sp.Ns);
                d2 = dot(-edge, next_vertex.sp.Ns);
                G_next = std::abs(d1 * d2) / edge_length2;
                next_vertex.a = curr_vertex.a * a;
                next_vertex.p_eye_A = q * p_psa * G_next;
                prev_vertex.p_light_A = q * p_psa * G_prev;
            }
        }
        return I;
    }

    /* uapp/emitter.cpp:19-30 with uapp/mesh.cpp:206-232 and uapp/implicitsphere.cpp:32-38 */
    void emitterRandomPoint(const Object& e, const Rng& rng, dvec3& pos, dvec3& Ng, dvec3& Ns, dvec3& Ts, dvec3& Bs) const {
        if (e.kind == UBDPT_MODEL_SPHERE) {
            Ns = Ng = sampleUnitSphereUniform(rng(DIM_POINT_A), rng(DIM_POINT_B));
            pos = Ng;
            Ts = -Ns + dvec3{0, 0, 1.0 / Ns.z};
            Bs = cross(Ns, Ts);
        } else {
            double r = rng(DIM_POINT_A);
            size_t f = 0;
            /* linear walk over the face probabilities; the reference falls off the end when rounding
             * leaves r >= 0 after the last face (uninitialised point) — here the last face is used */
            for (; f + 1 < e.num_faces; f++) {
                if (r < S.face_prob[e.first_face + f]) break;
                r -= S.face_prob[e.first_face + f];
            }
            const MeshVertex* face = &S.verts[3 * (e.first_face + f)];
            double u, v;
            pos = sampleTriangleUniform(face[0].pos, face[1].pos, face[2].pos, rng(DIM_POINT_B), rng(DIM_POINT_C), u, v);
            Ns = normalize((1.0 - u - v) * face[0].normal + u * face[1].normal + v * face[2].normal);
            Ng = normalize(cross(face[1].pos - face[0].pos, face[2].pos - face[0].pos));
            if (dot(Ns, Ng) < 0) Ng *= -1.0;
            Ts = normalize((1.0 - u - v) * face[0].tangent + u * face[1].tangent + v * face[2].tangent);
            Bs = normalize(cross(Ns, Ts));
        }
        pos += Ng * 0.0001;
        pos = transformPoint(e.W, pos);
        Ng = transformVectorT(e.invW, Ng);
        Ns = transformVectorT(e.invW, Ns);
        Ts = transformVectorT(e.invW, Ts);
        Bs = transformVectorT(e.invW, Bs);
    }

    /* ubdptrenderer.cpp:409-594 */
    void computeLightSubpath(std::vector<PathVertex>& subpath, const Rng& rng) const {
        subpath.clear();
        if (S.emitters.empty()) return; /* the reference dereferences a null emitter here */
        double p = rng(DIM_EMITTER);
        int ei = -1;
        for (int e : S.emitters) {
            if (p < S.objects[e].probability) { ei = e; break; }
            p -= S.objects[e].probability;
        }
        if (ei < 0) ei = S.emitters.back(); /* rounding fall-through: null dereference in the reference */
        const Object& emitter = S.objects[ei];

        dvec3 ePos, eNg, eNs, eTs, eBs;
        emitterRandomPoint(emitter, rng, ePos, eNg, eNs, eTs, eBs);

        PathVertex emitter_vertex{}; /* the reference leaves p_eye_A indeterminate (:433); 0 here */
        emitter_vertex.a = emitter.power;
        emitter_vertex.sp.object = ei;
        emitter_vertex.sp.pos = ePos;
        emitter_vertex.sp.Ng = eNg; emitter_vertex.sp.Ns = eNs; emitter_vertex.sp.Ts = eTs; emitter_vertex.sp.Bs = eBs;
        emitter_vertex.p_light_A = emitter.probability * (1.0 / emitter.area);
        emitter_vertex.specular = false;
        subpath.push_back(emitter_vertex);

        dvec3 dirT = samplePosHemUniform(rng(DIM_EMIT_ANGLE), rng(DIM_EMIT_V));
        fmat3 TNB{{to_f(eTs), to_f(eNs), to_f(eBs)}};
        dvec3 dirW = to_d(TNB * to_f(dirT));
        Ray ray(ePos, dirW);

        uint32_t h = 1;
        PathVertex next_vertex{};
        if (!S.intersectionPoint(ray, next_vertex.sp, nullptr, cn)) return;
        next_vertex.sp.lobe = S.pickLobe(next_vertex.sp.object, rng(hit_dim(h, DIM_LOBE)));
        if (next_vertex.sp.lobe == UBDPT_LOBE_NONE) return;

        /* QUIRK: next_vertex.sp.pos / Ns are object-local here (:463-469) */
        dvec3 edge = next_vertex.sp.pos - ePos;
        double edge_length2 = dot(edge, edge);
        edge /= std::sqrt(edge_length2);
        double d1 = dot(edge, eNs);
        double d2 = dot(-edge, next_vertex.sp.Ns);
        double G = std::abs(d1 * d2) / edge_length2;
        next_vertex.a = emitter_vertex.a;
        next_vertex.p_light_A = (1 / (2.0 * M_PI * dirT.y)) * G;

        while (true) {
            if (subpath.size() >= S.max_path_vertices) { if (cn) cn->capped++; break; } /* cap (not in reference) */
            const Object& obj = S.objects[next_vertex.sp.object];
            TNB = {{to_f(next_vertex.sp.Ts), to_f(next_vertex.sp.Ns), to_f(next_vertex.sp.Bs)}};
            BsdfInfo scatter_info = infoOf(next_vertex.sp);
            dvec3 w = transformVector(obj.invW, -ray.dir());
            double p_psa;
            dvec3 fs, next_dirT;
            dvec3 tex = texOf(next_vertex.sp);
            if (!bsdfScatter(next_vertex.sp.lobe, etaOf(next_vertex.sp), tex, scatter_info, w,
                             rng(hit_dim(h, DIM_SCATTER_1)), rng(hit_dim(h, DIM_SCATTER_2)), next_dirT, p_psa, fs,
                             next_vertex.specular))
                break;
            finishVertex(next_vertex, next_dirT);
            subpath.push_back(next_vertex);

            ray = Ray(subpath.back().sp.pos, transformVector(obj.W, to_d(TNB * to_f(next_dirT))));
            uint32_t h_curr = h;
            h++;
            if (!S.intersectionPoint(ray, next_vertex.sp, nullptr, cn)) break;
            next_vertex.sp.lobe = S.pickLobe(next_vertex.sp.object, rng(hit_dim(h, DIM_LOBE)));
            if (next_vertex.sp.lobe == UBDPT_LOBE_NONE) break;

            PathVertex& curr_vertex = subpath.back();
            PathVertex& prev_vertex = subpath[subpath.size() - 2];
            double fs_sum = fs.x + fs.y + fs.z;
            if (!(p_psa > 0) || !(fs_sum > 0)) break;
            double q;
            if (subpath.size() < S.min_depth) q = 1.0;
            else {
                q = std::min(1.0, (fs_sum / 3.0) / p_psa);
                if (rng(hit_dim(h_curr, DIM_ROULETTE)) > q) break;
            }
            if (curr_vertex.specular) {
                next_vertex.a = curr_vertex.a * fs;
                next_vertex.p_light_A = q * 1.0; /* QUIRK: eye path stores q*p_psa, light path q*1 (:557-558) */
                prev_vertex.p_eye_A = q * 1.0;
            } else {
                dvec3 a = fs / p_psa;
                double G_prev, G_next;
                edge = curr_vertex.sp.pos - prev_vertex.sp.pos;
                edge_length2 = dot(edge, edge);
                edge /= std::sqrt(edge_length2);
                d1 = dot(edge, prev_vertex.sp.Ns);
                d2 = dot(-edge, curr_vertex.sp.Ns);
                G_prev = std::abs(d1 * d2) / edge_length2;
                edge = next_vertex.sp.pos - curr_vertex.sp.pos; /* QUIRK: local-space next vertex (:578-582) */
                edge_length2 = dot(edge, edge);
                edge /= std::sqrt(edge_length2);
                d1 = dot(edge, curr_vertex.sp.Ns);
                d2 = dot(-edge, next_vertex.sp.Ns);
                G_next = std::abs(d1 * d2) / edge_length2;
                next_vertex.a = curr_vertex.a * a;
                next_vertex.p_light_A = q * p_psa * G_next;
                prev_vertex.p_eye_A = q * p_psa * G_prev;
            }
        }
    }

    /* ubdptrenderer.cpp:596-691 */
    bool connectionFactor(const std::vector<PathVertex>& L, const std::vector<PathVertex>& E, size_t s, size_t t,
                          dvec3& c) const {
        const PathVertex& vl = L[s - 1];
        const PathVertex& ve = E[t - 1];
        if (vl.specular || ve.specular) return false;
        if (!S.visibility(vl.sp.pos, ve.sp.pos, cn)) return false;
        dvec3 ce = vl.sp.pos - ve.sp.pos;
        double ce_length2 = dot(ce, ce);
        ce /= std::sqrt(ce_length2);
        double d1 = dot(ce, ve.sp.Ng);
        double d2 = dot(-ce, vl.sp.Ng);
        if ((d1 <= 0) || (d2 <= 0)) return false;
        d1 = dot(ce, ve.sp.Ns);
        d2 = dot(-ce, vl.sp.Ns);
        double G = (d1 * d2) / ce_length2;
        dvec3 fs1, fs2;
        if (s == 1)
            fs1 = dvec3{1, 1, 1} * (1.0 / (2.0 * M_PI * d2));
        else {
            dmat3 TNB{{vl.sp.Ts, vl.sp.Ns, vl.sp.Bs}};
            fs1 = bsdfSamplePSA(vl.sp.lobe, etaOf(vl.sp), texOf(vl.sp), infoOf(vl.sp),
                                normalize((L[s - 2].sp.pos - L[s - 1].sp.pos) * TNB),
                                normalize((E[t - 1].sp.pos - L[s - 1].sp.pos) * TNB));
            if (fs1.x + fs1.y + fs1.z <= 0) return false;
        }
        if (t == 1) {
            dvec3 edgeV = v3(S.V * v4(ce, 0));
            double d = S.image_plane_distance / edgeV.z;
            double d2_image_plane = dot(-edgeV, dvec3{0, 0, -1});
            double G_image_plane = std::abs(d1 * d2_image_plane) / (d * d);
            fs2 = dvec3{1, 1, 1} * ((1.0 / (S.image_plane_area)) / G_image_plane);
        } else {
            dmat3 TNB{{ve.sp.Ts, ve.sp.Ns, ve.sp.Bs}};
            fs2 = bsdfSamplePSA(ve.sp.lobe, etaOf(ve.sp), texOf(ve.sp), infoOf(ve.sp),
                                normalize((L[s - 1].sp.pos - E[t - 1].sp.pos) * TNB),
                                normalize((E[t - 2].sp.pos - E[t - 1].sp.pos) * TNB));
            if (fs2.x + fs2.y + fs2.z <= 0) return false;
        }
        c = fs1 * fs2 * G;
        return true;
    }

    /* ubdptrenderer.cpp:693-735 */
    double p_sp1(const std::vector<PathVertex>& L, const std::vector<PathVertex>& E, size_t s, size_t t) const {
        double ratio;
        const PathVertex& vl = L[s - 1];
        const PathVertex& ve = E[t - 1];
        dvec3 ce = vl.sp.pos - ve.sp.pos;
        double ce_length2 = dot(ce, ce);
        ce /= std::sqrt(ce_length2);
        double d1 = dot(ce, ve.sp.Ns);
        double d2 = dot(-ce, vl.sp.Ns);
        double G = (d1 * d2) / ce_length2;
        if (s == 1)
            ratio = (1.0 / (2.0 * M_PI * d2));
        else {
            dmat3 TNB{{vl.sp.Ts, vl.sp.Ns, vl.sp.Bs}};
            dvec3 wsT = normalize(E[t - 1].sp.pos - L[s - 1].sp.pos) * TNB;
            dvec3 wgT = normalize(L[s - 2].sp.pos - L[s - 1].sp.pos) * TNB;
            ratio = bsdfPPSA(vl.sp.lobe, etaOf(vl.sp), infoOf(vl.sp), wsT, wgT);
        }
        ratio *= G;
        return ratio / E[t - 1].p_eye_A;
    }
    /* ubdptrenderer.cpp:737-777 */
    double p_sm1(const std::vector<PathVertex>& L, const std::vector<PathVertex>& E, size_t s, size_t t) const {
        double p_eye_A;
        if (t == 1)
            p_eye_A = (1.0 / S.image_plane_area);
        else {
            const PathVertex& vl = L[s - 1];
            const PathVertex& ve = E[t - 1];
            dvec3 ce = vl.sp.pos - ve.sp.pos;
            double ce_length2 = dot(ce, ce);
            ce /= std::sqrt(ce_length2);
            double d1 = dot(ce, ve.sp.Ns);
            double d2 = dot(-ce, vl.sp.Ns);
            double G = (d1 * d2) / ce_length2;
            dmat3 TNB{{ve.sp.Ts, ve.sp.Ns, ve.sp.Bs}};
            dvec3 wsT = normalize(L[s - 1].sp.pos - E[t - 1].sp.pos) * TNB;
            dvec3 wgT = normalize(E[t - 2].sp.pos - E[t - 1].sp.pos) * TNB;
            p_eye_A = (bsdfPPSA(ve.sp.lobe, etaOf(ve.sp), infoOf(ve.sp), wsT, wgT) * G);
        }
        return (p_eye_A / L[s - 1].p_light_A);
    }
    /* ubdptrenderer.cpp:779-844: power heuristic, beta = 2 */
    double weight(const std::vector<PathVertex>& L, const std::vector<PathVertex>& E, size_t s, size_t t) const {
        double w = 1.0;
        double ratio = 1.0;
        if (t > 1) {
            ratio = p_sp1(L, E, s, t);
            if (!E[t - 2].specular) w += ratio * ratio;
        }
        if (t > 2)
            for (int v = (int)t - 2; v > 0; v--) {
                ratio *= E[v].p_light_A / E[v].p_eye_A;
                if (E[v].specular) continue;
                if (v > 0)
                    if (E[v - 1].specular) continue;
                w += ratio * ratio;
            }
        ratio = p_sm1(L, E, s, t);
        if (s > 1) {
            if (!L[s - 2].specular) w += ratio * ratio;
        } else
            w += ratio * ratio;
        if (s > 1)
            for (int v = (int)s - 2; v >= 0; v--) {
                ratio *= L[v].p_eye_A / L[v].p_light_A;
                if (L[v].specular) continue;
                if (v > 0)
                    if (L[v - 1].specular) continue;
                w += ratio * ratio;
            }
        return (1.0 / w);
    }

    /* ubdptrenderer.cpp:88-154. splat(x, y, value) receives the t == 1 contributions */
    template <class Splat>
    dvec3 renderPixel(size_t px, size_t py, uint64_t pass, uint64_t seed, std::vector<PathVertex>& eye_subpath,
                      std::vector<PathVertex>& light_subpath, Splat&& splat) const {
        dvec3 I{0, 0, 0};
        size_t pixel_sample = pass % S.num_pixel_strata;
        size_t lens_sample = pass % S.num_lens_strata;
        uint64_t key = path_key(seed, pass, (uint64_t)py * S.img_res_x + px);
        if (cn) cn->pixel_samples++;
        I += computeEyeSubpath(eye_subpath, px, py, lens_sample, pixel_sample, Rng{key, STREAM_EYE});
        computeLightSubpath(light_subpath, Rng{key, STREAM_LIGHT});
        if (cn) { cn->eye_v += eye_subpath.size(); cn->light_v += light_subpath.size(); }

        if ((light_subpath.size() > 0) && (eye_subpath.size() > 0))
            for (size_t s = 1; s <= light_subpath.size(); s++)
                for (size_t t = 1; t <= eye_subpath.size(); t++) {
                    size_t t1_pixel_x = 0, t1_pixel_y = 0;
                    if (t == 1) {
                        dvec3 rayW = light_subpath[s - 1].sp.pos - eye_subpath[0].sp.pos;
                        dvec3 rayV = normalize(v3(S.V * v4(rayW, 0)));
                        double d = S.image_plane_distance / rayV.z;
                        dvec3 lens_posV = v3(S.V * v4(eye_subpath[0].sp.pos, 1));
                        dvec3 ipV = lens_posV + d * rayV;
                        double pu = 0.5 * ((ipV.x / S.image_plane_ratio) + 1);
                        double pv = 1.0 - 0.5 * (ipV.y + 1);
                        /* the reference tests (pu<0)||(pu>1)||...; NaN would index out of range there */
                        if (!((pu >= 0) && (pu <= 1) && (pv >= 0) && (pv <= 1))) continue;
                        /* QUIRK: (W-1), (H-1) mapping (:129-130) */
                        t1_pixel_x = (size_t)std::floor((double)(S.img_res_x - 1) * pu);
                        t1_pixel_y = (size_t)std::floor((double)(S.img_res_y - 1) * pv);
                    }
                    dvec3 c;
                    if (!connectionFactor(light_subpath, eye_subpath, s, t, c)) continue;
                    double w = weight(light_subpath, eye_subpath, s, t);
                    if (cn) cn->conn++;
                    if (t == 1) {
                        dvec3 I_t1 = light_subpath[s - 1].a * eye_subpath[0].a * c * w;
                        if (cn) cn->splats++;
                        splat(t1_pixel_x, t1_pixel_y, I_t1);
                    } else
                        I += light_subpath[s - 1].a * eye_subpath[t - 1].a * c * w;
                }
        return I;
    }
};

inline void atomicAddDouble(double* p, double v) {
    uint64_t* ip = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        double cur;
        std::memcpy(&cur, &old, 8);
        double nv = cur + v;
        uint64_t nb;
        std::memcpy(&nb, &nv, 8);
        if (__atomic_compare_exchange_n(ip, &old, nb, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
    }
}

dmat4 mat_from(const double m[16]) {
    dmat4 r;
    std::memcpy(&r, m, sizeof(double) * 16);
    return r;
}
void mat_to(const dmat4& m, double out[16]) { std::memcpy(out, &m, sizeof(double) * 16); }

PathVertex vertex_from(const oracle_vertex& o) {
    auto get = [](const double* d) { return dvec3{d[0], d[1], d[2]}; };
    PathVertex v;
    v.sp.pos = get(o.pos); v.sp.Ng = get(o.Ng); v.sp.Ns = get(o.Ns); v.sp.Ts = get(o.Ts); v.sp.Bs = get(o.Bs);
    v.sp.tex_u = o.tex_u; v.sp.tex_v = o.tex_v;
    v.a = get(o.a);
    v.p_light_A = o.p_light_A; v.p_eye_A = o.p_eye_A;
    v.sp.object = o.object; v.sp.lobe = o.lobe; v.specular = o.specular != 0;
    return v;
}

void fill_vertex(const PathVertex& v, oracle_vertex& o) {
    auto put = [](double* d, dvec3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    put(o.pos, v.sp.pos); put(o.Ng, v.sp.Ng); put(o.Ns, v.sp.Ns); put(o.Ts, v.sp.Ts); put(o.Bs, v.sp.Bs);
    o.tex_u = v.sp.tex_u; o.tex_v = v.sp.tex_v;
    put(o.a, v.a);
    o.p_light_A = v.p_light_A; o.p_eye_A = v.p_eye_A;
    o.object = v.sp.object; o.lobe = v.sp.lobe; o.specular = v.specular ? 1 : 0; o.reserved = 0;
}

BsdfInfo info_from_query(const ubdpt_bsdf_query& q) {
    const double* f = q.frame;
    return {{f[0], f[1], f[2]}, {f[3], f[4], f[5]}, {f[6], f[7], f[8]}, {f[9], f[10], f[11]}, 1.0};
}

} // namespace

/* ============================================================================================ */
extern "C" {

int oracle_create(const ubdpt_scene_desc* sd, const ubdpt_render_params* rp, uint32_t max_path_vertices,
                  oracle_scene** out) {
    if (!sd || !rp || !out) return UBDPT_INVALID_ARG;
    if (rp->img_res_x == 0 || rp->img_res_y == 0 || rp->pixel_subdiv == 0 || rp->lens_subdiv == 0) return UBDPT_INVALID_ARG;
    oracle_scene* S = new oracle_scene();
    S->max_path_vertices = max_path_vertices ? max_path_vertices : 32;
    if (S->max_path_vertices < 2) S->max_path_vertices = 2;

    /* textures: uapp/textureimg.cpp:19-33 */
    S->textures.resize(sd->num_textures);
    for (uint32_t t = 0; t < sd->num_textures; t++) {
        const ubdpt_texture& in = sd->textures[t];
        Texture& tx = S->textures[t];
        tx.kind = in.kind; tx.width = in.width; tx.height = in.height;
        tx.color = {in.color[0], in.color[1], in.color[2]};
        if (in.kind == UBDPT_TEX_IMAGE) {
            size_t n = (size_t)in.width * in.height;
            if (n == 0 || in.texel_offset + 4 * n > sd->num_texel_bytes) { delete S; return UBDPT_INVALID_SCENE; }
            tx.pixels.resize(n);
            const uint8_t* src = sd->texels + in.texel_offset;
            for (size_t p = 0; p < n; p++)
                tx.pixels[p] = {(double)src[p * 4] / 255.0, (double)src[p * 4 + 1] / 255.0, (double)src[p * 4 + 2] / 255.0};
        }
    }
    /* mesh vertices: floats widened to double, uapp/mesh.cpp:14-32 */
    S->verts.resize(3 * sd->num_faces);
    for (size_t i = 0; i < 3 * sd->num_faces; i++) {
        const ubdpt_mesh_vertex& v = sd->vertices[i];
        MeshVertex& m = S->verts[i];
        m.pos = {v.pos[0], v.pos[1], v.pos[2]};
        m.normal = {v.normal[0], v.normal[1], v.normal[2]};
        m.tangent = {v.tangent[0], v.tangent[1], v.tangent[2]};
        m.tex_u = v.uv[0];
        m.tex_v = v.uv[1];
    }
    S->face_prob.assign(sd->num_faces, 0.0);
    S->objects.resize(sd->num_objects);
    for (uint32_t i = 0; i < sd->num_objects; i++) {
        const ubdpt_object& in = sd->objects[i];
        Object& o = S->objects[i];
        o.kind = in.kind; o.material = in.material; o.mat_a = in.mat_a; o.mat_b = in.mat_b;
        o.texture = in.texture; o.is_emitter = in.is_emitter != 0; o.emitter_index = -1;
        if (in.texture >= sd->num_textures) { delete S; return UBDPT_INVALID_SCENE; }
        o.W = mat_from(in.W);
        o.invW = inverse(o.W);
        o.power = {in.power[0], in.power[1], in.power[2]};
        o.first_face = in.first_face; o.num_faces = in.num_faces;
        o.area = 0; o.probability = 0;
        o.bounding_sphere = identity4();
        if (o.kind == UBDPT_MODEL_MESH) {
            if (o.num_faces == 0 || o.first_face + o.num_faces > sd->num_faces) { delete S; return UBDPT_INVALID_SCENE; }
            /* uapp/mesh.cpp:88-106 */
            double total_area = 0;
            std::vector<double> areas(o.num_faces);
            for (size_t f = 0; f < o.num_faces; f++) {
                const MeshVertex* fv = &S->verts[3 * (o.first_face + f)];
                double a = triangleArea(fv[0].pos, fv[1].pos, fv[2].pos);
                areas[f] = a;
                total_area += a;
            }
            for (size_t f = 0; f < o.num_faces; f++) S->face_prob[o.first_face + f] = areas[f] / total_area;
            o.bounding_sphere = S->computeBoundingSphere(o.first_face, o.num_faces);
        } else if (o.kind != UBDPT_MODEL_SPHERE) { delete S; return UBDPT_INVALID_SCENE; }
        if (o.is_emitter) {
            o.emitter_index = (int)S->emitters.size();
            S->emitters.push_back((int)i);
            if (o.kind == UBDPT_MODEL_SPHERE) {
                /* uapp/implicitsphere.cpp:25-30 */
                double R = length(v3(o.W * dvec4{1, 0, 0, 0}));
                o.area = 4.0 * M_PI * R * R;
            } else {
                /* uapp/mesh.cpp:191-204 */
                double A = 0;
                for (size_t f = 0; f < o.num_faces; f++) {
                    const MeshVertex* fv = &S->verts[3 * (o.first_face + f)];
                    A += triangleArea(v3(o.W * v4(fv[0].pos, 1.0)), v3(o.W * v4(fv[1].pos, 1.0)), v3(o.W * v4(fv[2].pos, 1.0)));
                }
                o.area = A;
            }
        }
    }
    /* uengine/uscene.cpp:4-19 */
    {
        double total = 0;
        for (int e : S->emitters) { dvec3 P = S->objects[e].power; total += (P.x + P.y + P.z) / S->objects[e].area; }
        for (int e : S->emitters) {
            dvec3 P = S->objects[e].power;
            S->objects[e].probability = (P.x + P.y + P.z) / (total * S->objects[e].area);
        }
    }
    /* uengine/ucamera.cpp:3-14,67-70 and uengine/ubdptrenderer.cpp:7-33 */
    const ubdpt_camera& cam = sd->camera;
    S->V = lookAtLH({cam.pos[0], cam.pos[1], cam.pos[2]}, {cam.look_at[0], cam.look_at[1], cam.look_at[2]}, {0, 1, 0});
    S->invV = inverse(S->V);
    S->image_plane_ratio = cam.ratio;
    S->image_plane_distance = 1.0 / tanf(cam.vfov / 2.0); /* QUIRK: float tan */
    S->img_res_x = rp->img_res_x; S->img_res_y = rp->img_res_y;
    S->num_pixel_strata = rp->pixel_subdiv * rp->pixel_subdiv;
    S->num_lens_strata = rp->lens_subdiv * rp->lens_subdiv;
    S->focus_plane_distance = rp->focus_plane_distance;
    S->lens_radius = rp->lens_size;
    S->min_depth = rp->min_depth;
    S->image_plane_area = 4 * S->image_plane_ratio;
    S->pixel_width = 2.0 * S->image_plane_ratio / (double)S->img_res_x;
    S->pixel_height = 2.0 / (double)S->img_res_y;
    S->lens_area = 0.0; /* QUIRK: never assigned (:29-30), value-initialised by make_unique (uengine.cpp:51) */
    *out = S;
    return UBDPT_SUCCESS;
}

void oracle_destroy(oracle_scene* s) { delete s; }

int oracle_set_accel(oracle_scene* s, int mode) {
    if (!s || (mode != 0 && mode != 1)) return UBDPT_INVALID_ARG;
    s->setAccel(mode);
    return UBDPT_SUCCESS;
}

int oracle_render_passes(oracle_scene* S, uint64_t first_pass, uint32_t n_passes, uint64_t pass_stride, uint64_t seed,
                         int num_threads, double* rgb, oracle_counters* counters) {
    if (!S || !rgb) return UBDPT_INVALID_ARG;
    if (num_threads < 1) num_threads = 1;
    if (pass_stride == 0) pass_stride = 1;
    const size_t W = S->img_res_x, H = S->img_res_y;
    std::vector<Counters> cns(num_threads);
    for (uint32_t i = 0; i < n_passes; i++) {
        uint64_t pass = first_pass + (uint64_t)i * pass_stride;
        std::atomic<size_t> next_row{0};
        auto work = [&](int tid) {
            Tracer tr(*S, counters ? &cns[tid] : nullptr);
            std::vector<PathVertex> E, L;
            for (;;) {
                size_t py = next_row.fetch_add(1);
                if (py >= H) break;
                for (size_t px = 0; px < W; px++) {
                    dvec3 I = tr.renderPixel(px, py, pass, seed, E, L, [&](size_t x, size_t y, dvec3 v) {
                        double* p = rgb + 3 * (y * W + x);
                        if (num_threads == 1) { p[0] += v.x; p[1] += v.y; p[2] += v.z; }
                        else { atomicAddDouble(p, v.x); atomicAddDouble(p + 1, v.y); atomicAddDouble(p + 2, v.z); }
                    });
                    double* p = rgb + 3 * (py * W + px);
                    if (num_threads == 1) { p[0] += I.x; p[1] += I.y; p[2] += I.z; }
                    else { atomicAddDouble(p, I.x); atomicAddDouble(p + 1, I.y); atomicAddDouble(p + 2, I.z); }
                }
            }
        };
        if (num_threads == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < num_threads; t++) th.emplace_back(work, t);
            for (auto& t : th) t.join();
        }
    }
    if (counters) {
        oracle_counters c{};
        for (auto& k : cns) {
            c.pixel_samples += k.pixel_samples; c.closest_hit_rays += k.closest; c.shadow_rays += k.shadow;
            c.eye_vertices += k.eye_v; c.light_vertices += k.light_v; c.connections += k.conn; c.splats += k.splats;
            c.capped_subpaths += k.capped; c.prims_tested += k.prims; c.nodes_visited += k.nodes;
            c.prims_tested_closest += k.prims_closest; c.nodes_visited_closest += k.nodes_closest;
        }
        *counters = c;
    }
    return UBDPT_SUCCESS;
}

int oracle_trace_pixel(oracle_scene* S, uint64_t pass, uint64_t seed, uint32_t px, uint32_t py, oracle_vertex* eye,
                       uint32_t* n_eye, oracle_vertex* light, uint32_t* n_light, double I[3], double* splats,
                       uint32_t max_splats, uint32_t* n_splats) {
    if (!S || px >= S->img_res_x || py >= S->img_res_y) return UBDPT_INVALID_ARG;
    Tracer tr(*S, nullptr);
    std::vector<PathVertex> E, L;
    uint32_t ns = 0;
    dvec3 Iv = tr.renderPixel(px, py, pass, seed, E, L, [&](size_t x, size_t y, dvec3 v) {
        if (splats && ns < max_splats) {
            double* p = splats + 5 * ns;
            p[0] = (double)x; p[1] = (double)y; p[2] = v.x; p[3] = v.y; p[4] = v.z;
        }
        ns++;
    });
    if (eye) for (size_t i = 0; i < E.size(); i++) fill_vertex(E[i], eye[i]);
    if (light) for (size_t i = 0; i < L.size(); i++) fill_vertex(L[i], light[i]);
    if (n_eye) *n_eye = (uint32_t)E.size();
    if (n_light) *n_light = (uint32_t)L.size();
    if (I) { I[0] = Iv.x; I[1] = Iv.y; I[2] = Iv.z; }
    if (n_splats) *n_splats = ns;
    return UBDPT_SUCCESS;
}

int oracle_connection_probe(oracle_scene* S, const oracle_vertex* eye, uint32_t n_eye, const oracle_vertex* light,
                            uint32_t n_light, ubdpt_connection* out, double* s0) {
    if (!S || !eye || !light || !out || n_eye == 0 || n_light == 0) return UBDPT_INVALID_ARG;
    Tracer tr(*S, nullptr);
    std::vector<PathVertex> E, L;
    for (uint32_t t = 0; t < n_eye; t++) E.push_back(vertex_from(eye[t]));
    for (uint32_t k = 0; k < n_light; k++) L.push_back(vertex_from(light[k]));
    for (uint32_t s = 1; s <= n_light; s++)
        for (uint32_t t = 1; t <= n_eye; t++) {
            ubdpt_connection& o = out[(size_t)(s - 1) * n_eye + (t - 1)];
            std::memset(&o, 0, sizeof(o));
            dvec3 c{0, 0, 0};
            if (!tr.connectionFactor(L, E, s, t, c)) continue;
            o.connected = 1;
            o.c[0] = c.x; o.c[1] = c.y; o.c[2] = c.z;
            o.w = tr.weight(L, E, s, t);
        }
    if (s0) {
        for (uint32_t k = 0; k < 3 * n_eye; k++) s0[k] = 0;
        for (uint32_t t = 1; t < n_eye; t++) {
            std::vector<PathVertex> sub(E.begin(), E.begin() + t);
            dvec3 v = tr.s0sample(sub, E[t]);
            s0[3 * t] = v.x; s0[3 * t + 1] = v.y; s0[3 * t + 2] = v.z;
        }
    }
    return UBDPT_SUCCESS;
}

int oracle_intersect_batch(oracle_scene* S, const ubdpt_ray* rays, size_t n, ubdpt_hit* hits) {
    if (!S || (!rays && n) || (!hits && n)) return UBDPT_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        Ray r({rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]}, {rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]});
        SurfacePoint sp;
        double d;
        ubdpt_hit& h = hits[i];
        std::memset(&h, 0, sizeof(h));
        if (S->intersectionPoint(r, sp, &d, nullptr)) {
            h.hit = 1; h.object = sp.object; h.face = sp.face; h.d = d;
            auto put = [](double* o, dvec3 s) { o[0] = s.x; o[1] = s.y; o[2] = s.z; };
            put(h.pos, sp.pos); put(h.Ng, sp.Ng); put(h.Ns, sp.Ns); put(h.Ts, sp.Ts); put(h.Bs, sp.Bs);
            h.tex_u = sp.tex_u; h.tex_v = sp.tex_v;
        } else {
            h.hit = 0; h.object = -1; h.face = -1; h.d = std::numeric_limits<double>::infinity();
        }
    }
    return UBDPT_SUCCESS;
}

int oracle_visibility_batch(oracle_scene* S, const ubdpt_segment* segs, size_t n, uint8_t* visible) {
    if (!S || (!segs && n) || (!visible && n)) return UBDPT_INVALID_ARG;
    for (size_t i = 0; i < n; i++)
        visible[i] = S->visibility({segs[i].p0[0], segs[i].p0[1], segs[i].p0[2]},
                                   {segs[i].p1[0], segs[i].p1[1], segs[i].p1[2]}, nullptr) ? 1 : 0;
    return UBDPT_SUCCESS;
}

int oracle_get_object_info(oracle_scene* S, uint32_t object, double invW[16], double* area, double* probability,
                           double bounding_sphere[16]) {
    if (!S || object >= S->objects.size()) return UBDPT_INVALID_ARG;
    const Object& o = S->objects[object];
    if (invW) mat_to(o.invW, invW);
    if (area) *area = o.area;
    if (probability) *probability = o.probability;
    if (bounding_sphere) mat_to(o.bounding_sphere, bounding_sphere);
    return UBDPT_SUCCESS;
}

int oracle_get_camera_info(oracle_scene* S, double view[16], double inv_view[16], double* image_plane_distance) {
    if (!S) return UBDPT_INVALID_ARG;
    if (view) mat_to(S->V, view);
    if (inv_view) mat_to(S->invV, inv_view);
    if (image_plane_distance) *image_plane_distance = S->image_plane_distance;
    return UBDPT_SUCCESS;
}

int oracle_bsdf_eval_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) {
    for (size_t i = 0; i < n; i++) {
        std::memset(&r[i], 0, sizeof(r[i]));
        dvec3 f = bsdfSamplePSA(q[i].lobe, q[i].eta, {q[i].tex[0], q[i].tex[1], q[i].tex[2]}, info_from_query(q[i]),
                                {q[i].wi[0], q[i].wi[1], q[i].wi[2]}, {q[i].wo[0], q[i].wo[1], q[i].wo[2]});
        r[i].f[0] = f.x; r[i].f[1] = f.y; r[i].f[2] = f.z;
    }
    return UBDPT_SUCCESS;
}
int oracle_bsdf_pdf_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) {
    for (size_t i = 0; i < n; i++) {
        std::memset(&r[i], 0, sizeof(r[i]));
        r[i].pdf = bsdfPPSA(q[i].lobe, q[i].eta, info_from_query(q[i]), {q[i].wi[0], q[i].wi[1], q[i].wi[2]},
                            {q[i].wo[0], q[i].wo[1], q[i].wo[2]});
    }
    return UBDPT_SUCCESS;
}
int oracle_bsdf_scatter_batch(const ubdpt_bsdf_query* q, size_t n, ubdpt_bsdf_result* r) {
    for (size_t i = 0; i < n; i++) {
        std::memset(&r[i], 0, sizeof(r[i]));
        dvec3 dir{0, 0, 0}, fs{0, 0, 0};
        double pdf = 0;
        bool spec = false;
        bool ok = bsdfScatter(q[i].lobe, q[i].eta, {q[i].tex[0], q[i].tex[1], q[i].tex[2]}, info_from_query(q[i]),
                              {q[i].wi[0], q[i].wi[1], q[i].wi[2]}, q[i].u1, q[i].u2, dir, pdf, fs, spec);
        r[i].ok = ok ? 1 : 0;
        if (ok) {
            r[i].f[0] = fs.x; r[i].f[1] = fs.y; r[i].f[2] = fs.z;
            r[i].pdf = pdf;
            r[i].dir[0] = dir.x; r[i].dir[1] = dir.y; r[i].dir[2] = dir.z;
            r[i].specular = spec ? 1 : 0;
        }
    }
    return UBDPT_SUCCESS;
}

int oracle_texture_sample_batch(oracle_scene* S, uint32_t tex, const double* uv, size_t n, double* rgb) {
    if (!S || tex >= S->textures.size()) return UBDPT_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        dvec3 c = S->textureSample(tex, uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
    return UBDPT_SUCCESS;
}

double oracle_uniform(uint64_t seed, uint64_t pass, uint64_t pixel, uint32_t stream, uint32_t dim) {
    return uniform01(path_key(seed, pass, pixel), stream, dim);
}

int oracle_sample_map(int which, uint64_t num_strata, uint64_t stratum, double u1, double u2, double out[3]) {
    out[0] = out[1] = out[2] = 0;
    switch (which) {
    case 0: { dvec2 r = sampleUnitRectStratified(num_strata, stratum, u1, u2); out[0] = r.x; out[1] = r.y; return 0; }
    case 1: { dvec2 r = sampleUnitDiskStratified(num_strata, stratum, u1, u2); out[0] = r.x; out[1] = r.y; return 0; }
    case 2: { dvec3 r = samplePosHemUniform(u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
    case 3: { dvec3 r = samplePosHemCos(u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
    case 4: { dvec3 r = sampleUnitSphereUniform(u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
    case 5: { double s = std::sqrt(u1), m = s * u2; out[0] = 1.0 - s; out[1] = s - m; out[2] = m; return 0; }
    }
    return UBDPT_INVALID_ARG;
}

/* ------------------------------------------------------------------------------------------ */
/* UEngine::imageRGB (uengine/uengine.cpp:182-202) with UConverter (uengine/uconverter.cpp)     */
/* ------------------------------------------------------------------------------------------ */
namespace {
/* uconverter.cpp:3-29: multi-lobe Gaussian fits of the CIE 1931 colour matching functions */
double xFit_1931(double wave) {
    double t1 = (wave - 442.0) * ((wave < 442.0) ? 0.0624 : 0.0374);
    double t2 = (wave - 599.8) * ((wave < 599.8) ? 0.0264 : 0.0323);
    double t3 = (wave - 501.1) * ((wave < 501.1) ? 0.0490 : 0.0382);
    return 0.362 * std::exp(-0.5 * t1 * t1) + 1.056 * std::exp(-0.5 * t2 * t2) - 0.065 * std::exp(-0.5 * t3 * t3);
}
double yFit_1931(double wave) {
    double t1 = (wave - 568.8) * ((wave < 568.8) ? 0.0213 : 0.0247);
    double t2 = (wave - 530.9) * ((wave < 530.9) ? 0.0613 : 0.0322);
    return 0.821 * std::exp(-0.5 * t1 * t1) + 0.286 * std::exp(-0.5 * t2 * t2);
}
double zFit_1931(double wave) {
    double t1 = (wave - 437.0) * ((wave < 437.0) ? 0.0845 : 0.0278);
    double t2 = (wave - 459.0) * ((wave < 459.0) ? 0.0385 : 0.0725);
    return 1.217 * std::exp(-0.5 * t1 * t1) + 0.681 * std::exp(-0.5 * t2 * t2);
}
/* uconverter.cpp:31-76: three 123 nm bins (r: 626.., g: 503.., b: 380..), 1 nm steps */
dvec3 xyzSum(double (*fit)(double)) {
    dvec3 sum = {0, 0, 0};
    for (size_t l = 0; l < 123; l++) {
        sum.x += fit(626 + l);
        sum.y += fit(503 + l);
        sum.z += fit(380 + l);
    }
    return sum;
}
/* uconverter.cpp:91-126 */
dvec3 radianceToSRGB(dvec3 radiance, double gamma) {
    static const dvec3 sumX = xyzSum(xFit_1931), sumY = xyzSum(yFit_1931), sumZ = xyzSum(zFit_1931);
    dvec3 xyz = {dot(radiance / 123.0, sumX), dot(radiance / 123.0, sumY), dot(radiance / 123.0, sumZ)};
    /* xyz * T with T's columns as written in the constructor (row vector times matrix) */
    dvec3 rgb = {dot(xyz, dvec3{3.2404542, -1.5371385, -0.4985314}), dot(xyz, dvec3{-0.9692660, 1.8760108, 0.0415560}),
                 dot(xyz, dvec3{0.0556434, -0.2040259, 1.0572252})};
    double c[3] = {rgb.x, rgb.y, rgb.z};
    for (int k = 0; k < 3; k++) {
        c[k] = std::fmin(std::fmax(c[k], 0.0), 1.0);
        if (c[k] <= 0.0031308) c[k] *= 12.92;
        else c[k] = 1.055 * std::pow(c[k], 1.0 / gamma) - 0.055;
        c[k] = std::fmin(std::fmax(c[k], 0.0), 1.0);
    }
    return {c[0], c[1], c[2]};
}
}  // namespace

int oracle_image_rgb(const double* sum, size_t npix, uint64_t curr_pass, double gamma, double* rgb) {
    if ((curr_pass == 0) || (gamma <= 0) || !sum || !rgb) return UBDPT_NO_DATA; /* uengine.cpp:184-185 */
    for (size_t p = 0; p < npix; p++) {
        dvec3 radiance = dvec3{sum[3 * p], sum[3 * p + 1], sum[3 * p + 2]} / static_cast<double>(curr_pass);
        dvec3 c = radianceToSRGB(radiance, gamma);
        rgb[3 * p] = c.x; rgb[3 * p + 1] = c.y; rgb[3 * p + 2] = c.z;
    }
    return UBDPT_SUCCESS;
}

} /* extern "C" */
