// scene_loader.h — scene ingestion for the HIP backend without Qt or assimp.
//
// Reads the reference's XML scene schema (uapp/scene.cpp:25-225: camera{ratio{w,h}, vfov, position,
// lookAt}, object[type=implicit_sphere|mesh]{texture | color, material(Glossy d,s | PerfectMirror |
// Dielectric eta | default LatexPaint), emit{r,g,b}, radius, center, file, translation,
// rotation{x,y,z,angle}, scale}) and produces the flat scene description of include/ubdpt.h with the
// objects in the order Scene::addObject / addEmitter would have pushed them.
//
// Meshes: Wavefront OBJ (v / vt / vn / f, polygons fan-triangulated), converted to the left-handed
// convention the reference asks assimp for (aiProcess_MakeLeftHanded: z negated), smooth normals
// generated when the file has none, per-vertex tangents from the uv gradients (Gram-Schmidt against
// the normal), float data like aiMesh. The reference's importer (assimp, :256-331) is not available
// here, so its exact vertex joining / tangent smoothing cannot be compared: same schema and
// conventions, not a bit-for-bit replacement of assimp.
// Image textures: binary PPM (P6); the reference decodes through QImage (uapp/textureimg.cpp:7-36).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "hip_renderer.h"

namespace uhip {

class LoadedScene : public FlatSceneSource {
public:
    // Parses `xml_path`; mesh and texture file names are resolved relative to `base_dir` (empty: as written).
    // On failure returns nullptr and sets `error`.
    static std::shared_ptr<LoadedScene> fromXml(const std::string& xml_path, const std::string& base_dir, std::string& error);
    static std::shared_ptr<LoadedScene> fromXmlText(const std::string& xml_text, const std::string& base_dir, std::string& error);

    bool flatten(ubdpt_scene_desc& out) override;
    const ubdpt_scene_desc& desc() const { return m_desc; }

private:
    void finalize();
    std::vector<ubdpt_object> m_objects;
    std::vector<ubdpt_texture> m_textures;
    std::vector<ubdpt_mesh_vertex> m_vertices;
    std::vector<uint8_t> m_texels;
    ubdpt_scene_desc m_desc{};
    friend struct SceneBuilderAccess;
};

}  // namespace uhip
