// asset_formats.h — readers for the mesh and image files a scene description names.
//
// The reference imports meshes through a third-party importer (any format it knows, uapp/scene.cpp:256-331) and images
// through QImage (any format it knows, uapp/textureimg.cpp:7-36). Neither library is part of the reference tree, so
// the readers here restate the documented file formats themselves: Wavefront OBJ (scene_loader.cpp), Stanford PLY
// (text and binary) and STL (text and binary) for meshes; Netpbm (scene_loader.cpp), PNG, BMP and TGA for images.
// JPEG is not read. All of them produce what the rest of the loader consumes: an indexed triangle soup per model
// and RGBA8888 rows from the top (QImage::Format_RGBA8888, textureimg.cpp:19-33).
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace uhip {
namespace formats {

struct V3 { double x, y, z; };
struct Corner { int p, t, n; };  // 1-based indices into pos / uv / nrm; 0 = the file has none

// What a mesh file holds before the importer's post-processing: positions, normals and texture coordinates as the
// file lists them (z already negated: the reference asks its importer for left-handed coordinates), and per model
// three corners per triangle.
struct MeshSoup {
    std::vector<V3> pos, nrm;
    std::vector<std::pair<double, double>> uv;
    std::vector<std::vector<Corner>> parts;
};

std::string lowerExtension(const std::string& path);  // "a/b/Mesh.PLY" -> "ply"

bool parsePly(const std::string& path, MeshSoup& out, std::string& error);
bool parseStl(const std::string& path, MeshSoup& out, std::string& error);

// PNG (all colour types and bit depths, interlaced or not), BMP (1 / 4 / 8 / 24 / 32 bits, uncompressed or
// bit fields) and TGA (true colour and grey, plain or run-length coded), chosen by the file's first bytes
// (TGA, which has none to go by, by its extension). Returns false when the file is none of them or damaged.
bool loadImageFile(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);

}  // namespace formats
}  // namespace uhip
