// scene.h — flattened scene as it lives in HBM, shared by host and device code.
//
// Layout decisions (MI355X): everything a ray touches on the way down the tree is 16-byte aligned
// and sized in whole 16-byte lanes (128-byte nodes, 48-byte primitives) so a lane fetches a node or a
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

// 128-byte four-wide BVH node (two 64-byte lines, fetched together): the boxes of up to four children
// (fp32, rounded outward and padded) and their links. Traversal is a chain of dependent node
// fetches, so halving the depth of the tree (the binary LBVH is collapsed two levels at a time)
// halves the number of memory round trips per ray.
// child >= 0: internal node index; child < 0: leaf holding primitive ~child; kEmptyChild: nothing.
struct __attribute__((aligned(16))) BvhNode {
    float lo[4][3];
    float hi[4][3];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");
constexpr int32_t kEmptyChild = 0x7fffffff;

// 48-byte primitive in tree (Morton) order: object-LOCAL float positions exactly as the mesh holds
// them (uapp/mesh.cpp:20-21), the owning object and the face index inside that object's mesh.
// Spheres are one primitive with face == kSphereFace.
struct __attribute__((aligned(16))) Prim {
    float p[9];
    uint32_t object;
    uint32_t face;
    uint32_t pad;
};
static_assert(sizeof(Prim) == 48, "Prim must be 48 bytes");
constexpr uint32_t kSphereFace = 0xffffffffu;

// Per-face shading attributes (normals, tangents, uv of the three corners), object order.
struct FaceAttr {
    float n[9];
    float t[9];
    float uv[6];
};
static_assert(sizeof(FaceAttr) == 96, "FaceAttr must be 96 bytes");

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
    const Prim* prims;
    uint32_t num_objects, num_emitters, num_prims, accel;
    uint32_t node_stride;  // bytes between two tree nodes: sizeof(BvhNode) in HBM, padded in an LDS copy
    uint32_t pad0;
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
