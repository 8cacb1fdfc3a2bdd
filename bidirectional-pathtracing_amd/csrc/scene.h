// scene.h — flattened scene as it lives in HBM, shared by host and device code.
//
// Layout decisions (MI355X): everything a ray touches on the way down the tree is 16-byte aligned
// and sized in whole 16-byte lanes (64-byte nodes, 64-byte primitives) so a lane fetches a node or a
// primitive with dwordx4 loads; per-face shading attributes are kept apart from positions because
// only the one winning face of a ray needs them.
#pragma once

#include <stdint.h>

#include "dmath.h"

namespace ub {

// One object of UScene::objects() (uapp/object.h:10-24, uapp/emitter.cpp:3-7).
struct DObject {
    m4 W;                 // object -> world
    m4 invW;              // uapp/object.cpp:8
    m4 bsphere;           // mesh: inverse(scale(translate(I, c), r)), uapp/mesh.cpp:85
    d3 power;             // emitter power
    double area;          // emitter area, uapp/emitter.cpp:6
    double prob;          // emitter pick probability, uengine/uscene.cpp:4-19
    double mat_a, mat_b;  // Glossy d, s | Dielectric eta
    uint32_t kind, material, texture, is_emitter;
    uint64_t first_face, num_faces;
    int32_t emitter_index;
    int32_t pad;
};

struct DTexture {
    uint32_t kind, width, height, pad;
    d3 color;
    uint64_t texel_offset;  // byte offset into the RGBA8888 texel pool
};

// Wide BVH node: the boxes of up to kNodeWidth children and their links, fetched together. The boxes are stored
// quantised: the node carries its own lower corner (fp32) and one power-of-two step per axis, a child box six
// bytes (lower corner rounded down, upper corner rounded up, in steps from the node's corner), so a four-wide
// node is one 64-byte line (four 16-byte loads per lane) and an eight-wide node 96 bytes. The builder checks that
// the box a lane decodes, with the very same fp32 operations, contains the exact one (bvh.hip), so traversal
// stays conservative and the result unchanged.
// Measured on a 2 M-triangle scene (closest-hit kernel, ms per launch): four-wide nodes of full fp32 boxes
// (128 bytes) 1.58, quantised four-wide 1.49, quantised eight-wide 1.52 with 40 % fewer node visits per ray
// (5.0 against 8.4): a wave spends as long in an eight-wide step as it saves in steps, so the narrower node with
// the smaller register footprint is the default.
// child >= 0: internal node index; child < 0: leaf holding primitive ~child; kEmptyChild: nothing.
#ifndef UBDPT_NODE_WIDTH
#define UBDPT_NODE_WIDTH 4
#endif
constexpr int kNodeWidth = UBDPT_NODE_WIDTH;
static_assert(kNodeWidth == 4 || kNodeWidth == 8, "node width: 4 or 8");
constexpr int kNodeQWords = kNodeWidth / 4;  // 32-bit words that hold one byte per child
struct __attribute__((aligned(16))) BvhNode {
    float origin[3];                 // lower corner of the union of the children
    uint32_t exps;                   // byte a: biased fp32 exponent of the step along axis a (0: step 0)
    int32_t child[kNodeWidth];
    uint32_t q[6][kNodeQWords];      // planes lo.x lo.y lo.z hi.x hi.y hi.z; byte c of a plane belongs to child c
    uint32_t pad[kNodeWidth == 4 ? 2 : 0];
};
static_assert(sizeof(BvhNode) == (kNodeWidth == 4 ? 64 : 96), "BvhNode is 64 or 96 bytes");
constexpr int32_t kEmptyChild = 0x7fffffff;
// The same node with full fp32 child boxes, built beside the quantised tree for scenes of at most kExactTreePrims
// primitives, whose tree is copied to LDS by the kernels that walk it: there node bytes cost nothing, and a room
// of axis-aligned walls is all thin leaves (6.1 node visits per ray through the quantised tree of a 14-primitive
// box, 2.5 through this one).
constexpr uint32_t kExactTreePrims = 48;
struct __attribute__((aligned(16))) ExactNode {
    float lo[kNodeWidth][3];
    float hi[kNodeWidth][3];
    int32_t child[kNodeWidth];
    int32_t pad[kNodeWidth == 4 ? 4 : 8];
};
static_assert(sizeof(ExactNode) == (kNodeWidth == 4 ? 128 : 256), "ExactNode is 128 or 256 bytes");
// step of a node along axis a, and the decoded coordinate of byte q: origin + q * step, one fp32 multiplication
// (exact) and one addition, never fused; builder and traversal both go through these two functions
__host__ __device__ inline float nodeStep(uint32_t exps, int a) {
    const uint32_t bits = ((exps >> (8 * a)) & 0xffu) << 23;
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}
__host__ __device__ inline float nodeCoord(float origin, uint32_t q, float step) {
    const float m = (float)q * step;
    return origin + m;
}

// Primitive in tree (Morton) order, one 64-byte line each (48-byte records straddle two lines half of the time;
// measured: 1 % apart): object-LOCAL float positions exactly as the mesh holds
// them (uapp/mesh.cpp:20-21), the owning object and the face index inside that object's mesh.
// Spheres are one primitive with face == kSphereFace.
#ifndef UBDPT_PRIM_BYTES
#define UBDPT_PRIM_BYTES 64
#endif
struct __attribute__((aligned(16))) Prim {
    float p[9];
    uint32_t object;
    uint32_t face;
    uint32_t pad[(UBDPT_PRIM_BYTES - 44) / 4];
};
static_assert(sizeof(Prim) == UBDPT_PRIM_BYTES && (UBDPT_PRIM_BYTES == 48 || UBDPT_PRIM_BYTES == 64), "Prim is 48 or 64 bytes");
constexpr uint32_t kSphereFace = 0xffffffffu;

// Per-face shading record, object order: normals, tangents, uv of the three corners, and the face's geometric normal
// normalize(cross(p1 - p0, p2 - p0)) of uapp/mesh.cpp:150, computed once from the float corners with the operations the
// kernels would use (before the flip towards the shading normal, which depends on the hit). One 128-byte line, the
// unit the memory system moves: shading a hit fetches this line and nothing else of the face (the corner positions and
// a 96-byte attribute record were three lines per hit, and a third of the shading kernel's time on a 2 M-triangle scene).
struct FaceAttr {
    float n[9];
    float t[9];
    float uv[6];
    double Ng[3];
    double pad;
};
static_assert(sizeof(FaceAttr) == 128, "FaceAttr must be one 128-byte line");

// Everything the kernels need, passed by value as a kernel argument.
struct DScene {
    const DObject* objects;
    const int32_t* emitters;  // object indices in UScene::emitters() order
    const float* face_pos;    // [num_faces][9], object order (linear-scan accelerator + winning-face reload)
    const FaceAttr* face_attr;
    const double* face_prob;  // uapp/mesh.cpp:88-106
    const double* face_cdf;   // per mesh: sum of face_prob over the faces before this one (running fp64 sum in face order)
    const DTexture* textures;
    const uint8_t* texels;
    const BvhNode* nodes;
    const ExactNode* xnodes;  // full-precision twin of `nodes` (small scenes only, else null); the LDS copy in kernels that stage the tree
    const Prim* prims;
    uint32_t num_objects, num_emitters, num_prims, accel;
    uint32_t node_stride;   // bytes between two tree nodes: sizeof(BvhNode) in HBM
    uint32_t xnode_stride;  // the same for xnodes: sizeof(ExactNode) in HBM, padded in an LDS copy
    uint64_t num_faces;

    // UBDPTRenderer members (uengine/ubdptrenderer.h:54-79)
    m4 V, invV;
    double image_plane_ratio, image_plane_distance, image_plane_area;
    double pixel_width, pixel_height;
    double focus_plane_distance, lens_radius, lens_area;
    uint32_t img_res_x, img_res_y;
    uint32_t num_pixel_strata, num_lens_strata;
    uint32_t min_depth, max_path_vertices;
};

}  // namespace ub
