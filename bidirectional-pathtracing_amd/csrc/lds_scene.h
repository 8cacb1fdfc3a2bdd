// lds_scene.h — copies of small scenes in LDS (included by render.hip before the wavefront kernels).
//
// The kernels read parts of a hit's object record (transforms, material, emitter data: 480 bytes per
// object) at several points, each read depending on the one before, and a ray walks the tree as a
// chain of dependent node and primitive fetches. From HBM / L2 every link of such a chain costs
// hundreds of cycles; from LDS tens. Scenes with few objects therefore get their object table
// copied to LDS by every workgroup, and scenes of a few dozen primitives their tree as well. The
// kernels are instantiated per placement (a pointer that may point to either memory would have to
// be dereferenced with the slower flat instructions), and the launchers pick the instantiation.
#pragma once

namespace ub {

constexpr uint32_t kLdsObjectsShade = 64;  // shading kernels hold 2 workgroups per CU
#ifndef UBDPT_LDS_OBJECTS_TRACE
#define UBDPT_LDS_OBJECTS_TRACE 16
#endif
// ray tracing kernels hold 4 workgroups per CU and keep their traversal stacks in LDS as well; with room for 32 objects
// (15 KB) the box of 14 primitives rendered 2 to 3 % slower and the 2 M-triangle scene the same
constexpr uint32_t kLdsObjectsTrace = UBDPT_LDS_OBJECTS_TRACE;
constexpr uint32_t kLdsTreePrims = kExactTreePrims;

// Every thread of the workgroup must call this; the returned pointer replaces DScene::objects.
DEV const DObject* stageObjects(const DScene& S, DObject* lds) {
    static_assert(sizeof(DObject) % 16 == 0, "DObject is copied in 16-byte words");
    const uint4* src = reinterpret_cast<const uint4*>(S.objects);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    const uint32_t words = S.num_objects * (uint32_t)(sizeof(DObject) / 16);
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) dst[w] = src[w];
    __syncthreads();
    return lds;
}

// In LDS a node is followed by 16 bytes of padding: with the power-of-two stride of the HBM layout the lanes of
// a wave that read the same field of different nodes would all hit the same few banks.
struct __attribute__((aligned(16))) LdsNode {
    ExactNode n;
    uint32_t pad[4];
};

// Every thread of the workgroup must call this; S.xnodes and S.prims then point at the LDS copies of the
// full-precision tree and the primitives (a kernel that calls this walks the tree with EXACT = true).
DEV void stageTree(DScene& S, LdsNode* lds_nodes, Prim* lds_prims) {
    static_assert(sizeof(ExactNode) % 16 == 0 && sizeof(LdsNode) % 16 == 0 && sizeof(Prim) % 16 == 0, "copied in 16-byte words");
    constexpr uint32_t kNodeWords = sizeof(ExactNode) / 16, kLdsNodeWords = sizeof(LdsNode) / 16;
    const uint32_t n_nodes = S.num_prims > 1 ? S.num_prims - 1 : 1;  // slots of the node array: one per node of the binary tree
    const uint4* src = reinterpret_cast<const uint4*>(S.xnodes);
    uint4* dst = reinterpret_cast<uint4*>(lds_nodes);
    for (uint32_t w = threadIdx.x; w < n_nodes * kNodeWords; w += kBlock)
        dst[(w / kNodeWords) * kLdsNodeWords + (w % kNodeWords)] = src[w];
    src = reinterpret_cast<const uint4*>(S.prims);
    dst = reinterpret_cast<uint4*>(lds_prims);
    for (uint32_t w = threadIdx.x; w < S.num_prims * (uint32_t)(sizeof(Prim) / 16); w += kBlock) dst[w] = src[w];
    __syncthreads();
    S.xnodes = &lds_nodes[0].n;
    S.xnode_stride = (uint32_t)sizeof(LdsNode);
    S.prims = lds_prims;
}

// LDS_LEVEL of the kernels that trace rays: 0 = scene in HBM, 1 = object table in LDS,
// 2 = object table, tree nodes and primitives in LDS
DEV constexpr uint32_t ldsObjects(int level) { return level >= 1 ? kLdsObjectsTrace : 1; }
DEV constexpr uint32_t ldsPrims(int level) { return level >= 2 ? kLdsTreePrims : 1; }
DEV constexpr uint32_t ldsNodes(int level) { return level >= 2 ? kLdsTreePrims : 1; }
inline int ldsLevelFor(const DScene& S) {
    if (S.num_objects > kLdsObjectsTrace) return 0;
    return (S.accel == 0 && S.xnodes && S.num_prims <= kLdsTreePrims) ? 2 : 1;
}

}  // namespace ub
