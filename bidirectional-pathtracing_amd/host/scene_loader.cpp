// scene_loader.cpp — see scene_loader.h.
#include "scene_loader.h"

#include <new>
#include <exception>

#include "asset_formats.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace uhip {

namespace {

// ---- a small XML reader: elements, attributes, text; comments, declarations and CDATA are skipped ------
struct XmlNode {
    std::string tag, text;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<XmlNode>> children;

    // QDomElement::elementsByTagName: every DESCENDANT with that tag, document order
    void collect(const std::string& name, std::vector<const XmlNode*>& out) const {
        for (const auto& c : children) {
            if (c->tag == name) out.push_back(c.get());
            c->collect(name, out);
        }
    }
    std::vector<const XmlNode*> byTag(const std::string& name) const {
        std::vector<const XmlNode*> v;
        collect(name, v);
        return v;
    }
    const XmlNode* first(const std::string& name) const {
        auto v = byTag(name);
        return v.empty() ? nullptr : v[0];
    }
};

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

class XmlParser {
public:
    explicit XmlParser(const std::string& s) : m_s(s) {}
    std::unique_ptr<XmlNode> parse(std::string& error) {
        auto root = std::make_unique<XmlNode>();
        if (!parseChildren(*root, "", error)) return nullptr;
        if (root->children.empty()) { error = "no root element"; return nullptr; }
        return root;
    }

private:
    bool skipMisc() {  // comments, processing instructions, doctype
        for (;;) {
            if (m_s.compare(m_p, 4, "<!--") == 0) {
                size_t e = m_s.find("-->", m_p);
                if (e == std::string::npos) return false;
                m_p = e + 3;
            } else if (m_s.compare(m_p, 2, "<?") == 0) {
                size_t e = m_s.find("?>", m_p);
                if (e == std::string::npos) return false;
                m_p = e + 2;
            } else if (m_s.compare(m_p, 2, "<!") == 0) {
                size_t e = m_s.find('>', m_p);
                if (e == std::string::npos) return false;
                m_p = e + 1;
            } else
                return true;
        }
    }
    bool parseChildren(XmlNode& parent, const std::string& closing, std::string& error) {
        for (;;) {
            size_t lt = m_s.find('<', m_p);
            if (lt == std::string::npos) {
                if (!closing.empty()) { error = "unterminated element <" + closing + ">"; return false; }
                return true;
            }
            parent.text += m_s.substr(m_p, lt - m_p);
            m_p = lt;
            if (m_s.compare(m_p, 2, "<!") == 0 || m_s.compare(m_p, 2, "<?") == 0) {
                if (!skipMisc()) { error = "unterminated comment or declaration"; return false; }
                continue;
            }
            if (m_s.compare(m_p, 2, "</") == 0) {
                size_t gt = m_s.find('>', m_p);
                if (gt == std::string::npos) { error = "unterminated closing tag"; return false; }
                std::string name = trim(m_s.substr(m_p + 2, gt - m_p - 2));
                m_p = gt + 1;
                if (name != closing) { error = "mismatched </" + name + ">"; return false; }
                parent.text = trim(parent.text);
                return true;
            }
            // opening tag
            size_t gt = m_s.find('>', m_p);
            if (gt == std::string::npos) { error = "unterminated tag"; return false; }
            bool self = gt > 0 && m_s[gt - 1] == '/';
            std::string inside = m_s.substr(m_p + 1, gt - m_p - 1 - (self ? 1 : 0));
            m_p = gt + 1;
            auto node = std::make_unique<XmlNode>();
            size_t k = 0;
            while (k < inside.size() && !isspace((unsigned char)inside[k])) k++;
            node->tag = inside.substr(0, k);
            // attributes: name = "value" | 'value'
            while (k < inside.size()) {
                while (k < inside.size() && isspace((unsigned char)inside[k])) k++;
                size_t eq = inside.find('=', k);
                if (eq == std::string::npos) break;
                std::string an = trim(inside.substr(k, eq - k));
                size_t q = inside.find_first_of("\"'", eq);
                if (q == std::string::npos) break;
                size_t q2 = inside.find(inside[q], q + 1);
                if (q2 == std::string::npos) break;
                node->attr[an] = inside.substr(q + 1, q2 - q - 1);
                k = q2 + 1;
            }
            if (!self && !parseChildren(*node, node->tag, error)) return false;
            parent.children.push_back(std::move(node));
        }
    }
    const std::string& m_s;
    size_t m_p = 0;
};

// QString::toDouble of a missing node / unparsable text is 0.0
double num(const XmlNode* n) { return n ? std::strtod(n->text.c_str(), nullptr) : 0.0; }
double child(const XmlNode* n, const char* tag) { return n ? num(n->first(tag)) : 0.0; }
double attrNum(const XmlNode* n, const char* name) {
    if (!n) return 0.0;
    auto it = n->attr.find(name);
    return it == n->attr.end() ? 0.0 : std::strtod(it->second.c_str(), nullptr);
}

// ---- column-major 4x4, the few GLM calls of uapp/scene.cpp:137,179 ------------------------------------
struct M4 { double c[4][4]; };  // c[column][row]
M4 identity() {
    M4 m{};
    for (int i = 0; i < 4; i++) m.c[i][i] = 1.0;
    return m;
}
M4 translate(const M4& m, const double t[3]) {  // glm::translate: m[3] = m[0]*t0 + m[1]*t1 + m[2]*t2 + m[3]
    M4 r = m;
    for (int k = 0; k < 4; k++) r.c[3][k] = m.c[0][k] * t[0] + m.c[1][k] * t[1] + m.c[2][k] * t[2] + m.c[3][k];
    return r;
}
M4 scale(const M4& m, const double s[3]) {
    M4 r = m;
    for (int k = 0; k < 4; k++) { r.c[0][k] = m.c[0][k] * s[0]; r.c[1][k] = m.c[1][k] * s[1]; r.c[2][k] = m.c[2][k] * s[2]; }
    return r;
}
M4 rotate(const M4& m, double angle, const double v[3]) {  // glm::rotate(m, angle, v)
    const double c = std::cos(angle), s = std::sin(angle);
    double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double a[3] = {v[0] * (1.0 / len), v[1] * (1.0 / len), v[2] * (1.0 / len)};
    double t[3] = {(1.0 - c) * a[0], (1.0 - c) * a[1], (1.0 - c) * a[2]};
    double R[3][3];
    R[0][0] = c + t[0] * a[0]; R[0][1] = t[0] * a[1] + s * a[2]; R[0][2] = t[0] * a[2] - s * a[1];
    R[1][0] = t[1] * a[0] - s * a[2]; R[1][1] = c + t[1] * a[1]; R[1][2] = t[1] * a[2] + s * a[0];
    R[2][0] = t[2] * a[0] + s * a[1]; R[2][1] = t[2] * a[1] - s * a[0]; R[2][2] = c + t[2] * a[2];
    M4 r = m;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++) r.c[j][k] = m.c[0][k] * R[j][0] + m.c[1][k] * R[j][1] + m.c[2][k] * R[j][2];
    return r;
}

// ---- Wavefront OBJ -> triangle soup in the conventions the reference requests from its importer ------
using formats::Corner;
using formats::MeshSoup;
using formats::V3;
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 mul(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 normalized(V3 a) {
    double l = std::sqrt(dot(a, a));
    return l > 0 ? mul(a, 1.0 / l) : V3{0, 0, 0};
}

// Reads a Wavefront OBJ file: positions, normals and texture coordinates as listed, and per model three corners per
// triangle. The reference hands every aiMesh of the imported file to its own
// Object/Emitter (uapp/scene.cpp:181-190, :284-327); the importer makes one mesh per `o` object and material
// and, with aiProcess_OptimizeMeshes, joins the `g` groups of one object that share a material. The importer
// itself is a third-party library that is not part of the reference tree, so this division (and the order of
// the parts: first appearance in the file) restates its documented behaviour and is not pinned by a fixture.
bool parseObj(const std::string& path, MeshSoup& soup, std::string& error) {
    std::ifstream in(path);
    if (!in.is_open()) { error = "cannot open mesh file " + path; return false; }
    std::vector<V3>& pos = soup.pos;
    std::vector<V3>& nrm = soup.nrm;
    std::vector<std::pair<double, double>>& uv = soup.uv;
    std::vector<std::vector<Corner>>& parts = soup.parts;  // per model: 3 corners per triangle
    std::map<std::pair<int, std::string>, size_t> part_of;
    int object_index = 0;
    std::string material;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string kw;
        ls >> kw;
        if (kw == "o") object_index++;
        else if (kw == "usemtl") { material.clear(); ls >> material; }
        else if (kw == "v") { V3 p{}; ls >> p.x >> p.y >> p.z; p.z = -p.z; pos.push_back(p); }         // left-handed: z negated
        else if (kw == "vn") { V3 n{}; ls >> n.x >> n.y >> n.z; n.z = -n.z; nrm.push_back(n); }
        else if (kw == "vt") { double u = 0, v = 0; ls >> u >> v; uv.emplace_back(u, v); }
        else if (kw == "f") {
            std::vector<Corner> poly;
            std::string tok;
            while (ls >> tok) {
                Corner c{0, 0, 0};
                size_t s1 = tok.find('/');
                c.p = std::atoi(tok.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) c.t = std::atoi(t.c_str());
                    if (s2 != std::string::npos) c.n = std::atoi(tok.substr(s2 + 1).c_str());
                }
                // negative indices count from the end
                if (c.p < 0) c.p = (int)pos.size() + c.p + 1;
                if (c.t < 0) c.t = (int)uv.size() + c.t + 1;
                if (c.n < 0) c.n = (int)nrm.size() + c.n + 1;
                poly.push_back(c);
            }
            if (poly.size() < 3) continue;  // points and lines: aiProcess_SortByPType keeps them out of triangle meshes
            auto key = std::make_pair(object_index, material);
            auto it = part_of.find(key);
            if (it == part_of.end()) { it = part_of.emplace(key, parts.size()).first; parts.emplace_back(); }
            std::vector<Corner>& corners = parts[it->second];
            for (size_t k = 1; k + 1 < poly.size(); k++) {  // aiProcess_Triangulate: fan
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
            }
        }
    }
    if (parts.empty()) { error = "mesh has no faces: " + path; return false; }
    return true;
}

// The importer's post-processing of a mesh file's models, restated: triangle corners appended to `out`, `part_faces`
// receives the face count of each model in order.
bool emitModels(const std::string& path, const MeshSoup& soup, std::vector<ubdpt_mesh_vertex>& out,
                std::vector<size_t>& part_faces, std::string& error) {
    const std::vector<V3>& pos = soup.pos;
    const std::vector<V3>& nrm = soup.nrm;
    const std::vector<std::pair<double, double>>& uv = soup.uv;
    for (const std::vector<Corner>& corners : soup.parts)
        for (const Corner& c : corners)
            if (c.p < 1 || c.p > (int)pos.size() || c.t < 0 || c.t > (int)uv.size() || c.n < 0 || c.n > (int)nrm.size()) {
                error = "face index out of range in " + path;
                return false;
            }
    for (const std::vector<Corner>& corners : soup.parts) {
        // aiProcess_GenSmoothNormals when the file has none: area-weighted face normals summed per position,
        // within one model (the importer smooths each mesh on its own)
        std::vector<V3> smooth;
        bool need_normals = false;
        for (const Corner& c : corners) need_normals |= (c.n == 0);
        if (need_normals) {
            smooth.assign(pos.size(), V3{0, 0, 0});
            for (size_t f = 0; f < corners.size(); f += 3) {
                V3 p0 = pos[corners[f].p - 1], p1 = pos[corners[f + 1].p - 1], p2 = pos[corners[f + 2].p - 1];
                V3 fn = cross(sub(p1, p0), sub(p2, p0));
                for (int k = 0; k < 3; k++) smooth[corners[f + k].p - 1] = add(smooth[corners[f + k].p - 1], fn);
            }
        }
        out.reserve(out.size() + corners.size());
        for (size_t f = 0; f < corners.size(); f += 3) {
            V3 p[3], n[3];
            double u[3], v[3];
            for (int k = 0; k < 3; k++) {
                const Corner& c = corners[f + k];
                p[k] = pos[c.p - 1];
                n[k] = normalized(c.n ? nrm[c.n - 1] : smooth[c.p - 1]);
                // aiProcess_GenUVCoords without a mapping leaves missing coordinates at 0
                u[k] = c.t ? uv[c.t - 1].first : 0.0;
                v[k] = c.t ? uv[c.t - 1].second : 0.0;
            }
            // aiProcess_CalcTangentSpace, per face: tangent along increasing u
            V3 e1 = sub(p[1], p[0]), e2 = sub(p[2], p[0]);
            double sx = u[1] - u[0], sy = v[1] - v[0], tx = u[2] - u[0], ty = v[2] - v[0];
            double dir = (tx * sy - ty * sx) < 0.0 ? -1.0 : 1.0;
            if (sx * ty == sy * tx) { sx = 0.0; sy = 1.0; tx = 1.0; ty = 0.0; }
            V3 tangent = mul(sub(mul(e2, sy), mul(e1, ty)), dir);
            for (int k = 0; k < 3; k++) {
                V3 t = normalized(sub(tangent, mul(n[k], dot(tangent, n[k]))));  // Gram-Schmidt against the vertex normal
                if (t.x == 0 && t.y == 0 && t.z == 0) {  // degenerate uv: any vector orthogonal to the normal
                    V3 h = std::fabs(n[k].x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
                    t = normalized(cross(h, n[k]));
                }
                ubdpt_mesh_vertex mv;
                mv.pos[0] = (float)p[k].x; mv.pos[1] = (float)p[k].y; mv.pos[2] = (float)p[k].z;
                mv.normal[0] = (float)n[k].x; mv.normal[1] = (float)n[k].y; mv.normal[2] = (float)n[k].z;
                mv.tangent[0] = (float)t.x; mv.tangent[1] = (float)t.y; mv.tangent[2] = (float)t.z;
                mv.uv[0] = (float)u[k]; mv.uv[1] = (float)v[k];
                out.push_back(mv);
            }
        }
        part_faces.push_back(corners.size() / 3);
    }
    return true;
}

// A mesh file by its extension: Wavefront OBJ, Stanford PLY, STL. (The reference's importer knows more formats; these
// are the ones restated here, asset_formats.h.)
bool loadMesh(const std::string& path, std::vector<ubdpt_mesh_vertex>& out, std::vector<size_t>& part_faces,
              std::string& error) {
    MeshSoup soup;
    const std::string ext = formats::lowerExtension(path);
    bool ok;
    if (ext == "ply") ok = formats::parsePly(path, soup, error);
    else if (ext == "stl") ok = formats::parseStl(path, soup, error);
    else ok = parseObj(path, soup, error);
    return ok && emitModels(path, soup, out, part_faces, error);
}

// An image file as RGBA8888 rows from the top: Netpbm here, PNG / BMP / TGA in asset_formats.cpp.
bool loadPpm(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
bool loadTexture(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    return loadPpm(path, w, h, rgba) || formats::loadImageFile(path, w, h, rgba);
}

// Netpbm images with 8-bit samples: P6 / P3 (colour, binary / text) and P5 / P2 (grey, binary / text), as RGBA8888
// (what TextureImg keeps of a QImage, uapp/textureimg.cpp:19-33).
bool loadPpm(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) return false;
    std::string magic;
    in >> magic;
    if (magic != "P6" && magic != "P3" && magic != "P5" && magic != "P2") return false;
    const bool text = magic == "P3" || magic == "P2";
    const size_t channels = (magic == "P6" || magic == "P3") ? 3 : 1;
    auto nextInt = [&](long& v) {
        for (;;) {
            int c = in.peek();
            if (c == '#') { std::string skip; std::getline(in, skip); }
            else if (isspace(c)) in.get();
            else break;
        }
        return (bool)(in >> v);
    };
    long W = 0, H = 0, maxv = 0;
    if (!nextInt(W) || !nextInt(H) || !nextInt(maxv) || W <= 0 || H <= 0 || maxv != 255) return false;
    std::vector<uint8_t> samples((size_t)W * H * channels);
    if (text) {
        for (uint8_t& x : samples) {
            long v = 0;
            if (!nextInt(v) || v < 0 || v > 255) return false;
            x = (uint8_t)v;
        }
    } else {
        in.get();  // the single whitespace byte after maxval
        in.read(reinterpret_cast<char*>(samples.data()), (std::streamsize)samples.size());
        if ((size_t)in.gcount() != samples.size()) return false;
    }
    w = (uint32_t)W; h = (uint32_t)H;
    rgba.resize((size_t)W * H * 4);
    for (size_t p = 0; p < (size_t)W * H; p++) {
        for (size_t k = 0; k < 3; k++) rgba[4 * p + k] = samples[channels * p + (channels == 3 ? k : 0)];
        rgba[4 * p + 3] = 255;
    }
    return true;
}

std::string joinPath(const std::string& base, const std::string& name) {
    if (base.empty() || (!name.empty() && name[0] == '/')) return name;
    return base.back() == '/' ? base + name : base + "/" + name;
}

}  // namespace

struct SceneBuilderAccess {
    static uint32_t colorTexture(LoadedScene& s, double r, double g, double b) {
        ubdpt_texture t{};
        t.kind = UBDPT_TEX_COLOR;
        t.color[0] = r; t.color[1] = g; t.color[2] = b;
        s.m_textures.push_back(t);
        return (uint32_t)s.m_textures.size() - 1;
    }

    // uapp/scene.cpp:57-192
    static bool object(LoadedScene& s, const XmlNode& xo, const std::string& base_dir,
                       std::map<std::string, uint32_t>& image_cache, std::string& error) {
        // texture: image file, colour, or the default grey (:60-83)
        uint32_t tex;
        if (const XmlNode* xt = xo.first("texture")) {
            std::string file = xt->text;
            auto it = image_cache.find(file);  // TextureImg::get caches by path name (textureimg.cpp:38-53)
            if (it != image_cache.end()) tex = it->second;
            else {
                uint32_t w = 0, h = 0;
                std::vector<uint8_t> rgba;
                if (loadTexture(joinPath(base_dir, file), w, h, rgba)) {
                    ubdpt_texture t{};
                    t.kind = UBDPT_TEX_IMAGE; t.width = w; t.height = h; t.texel_offset = s.m_texels.size();
                    s.m_texels.insert(s.m_texels.end(), rgba.begin(), rgba.end());
                    s.m_textures.push_back(t);
                    tex = (uint32_t)s.m_textures.size() - 1;
                    image_cache[file] = tex;
                } else {
                    error = "Failed to load texture from file (" + file + ")";
                    return false;
                }
            }
        } else if (const XmlNode* xc = xo.first("color")) {
            tex = colorTexture(s, child(xc, "r"), child(xc, "g"), child(xc, "b"));
        } else
            tex = colorTexture(s, 0.8, 0.8, 0.8);

        // material (:86-101): the element's text selects the class, parameters are attributes
        ubdpt_object proto{};
        proto.material = UBDPT_MAT_LATEX_PAINT;
        proto.texture = tex;
        const XmlNode* xm = xo.first("material");
        std::string mat = xm ? xm->text : std::string();
        if (mat == "Glossy") { proto.material = UBDPT_MAT_GLOSSY; proto.mat_a = attrNum(xm, "d"); proto.mat_b = attrNum(xm, "s"); }
        else if (mat == "PerfectMirror") proto.material = UBDPT_MAT_PERFECT_MIRROR;
        else if (mat == "Dielectric") { proto.material = UBDPT_MAT_DIELECTRIC; proto.mat_a = attrNum(xm, "eta"); }

        // emission (:104-114)
        if (const XmlNode* xe = xo.first("emit")) {
            proto.is_emitter = 1;
            proto.power[0] = child(xe, "r"); proto.power[1] = child(xe, "g"); proto.power[2] = child(xe, "b");
        }

        auto it = xo.attr.find("type");
        std::string type = it == xo.attr.end() ? std::string() : it->second;
        if (type == "implicit_sphere") {  // :121-144
            double center[3] = {0, 0, 0}, radius = 1.0;
            if (const XmlNode* xr = xo.first("radius")) radius = num(xr);
            if (const XmlNode* xc = xo.first("center")) { center[0] = child(xc, "x"); center[1] = child(xc, "y"); center[2] = child(xc, "z"); }
            double rr[3] = {radius, radius, radius};
            M4 T = scale(translate(identity(), center), rr);
            proto.kind = UBDPT_MODEL_SPHERE;
            std::memcpy(proto.W, T.c, sizeof(proto.W));
            s.m_objects.push_back(proto);
        } else if (type == "mesh") {  // :145-191
            const XmlNode* xf = xo.first("file");
            double trans[3] = {0, 0, 0}, rot[3] = {1, 1, 1}, scl[3] = {1, 1, 1}, angle = 0;
            if (const XmlNode* x = xo.first("translation")) { trans[0] = child(x, "x"); trans[1] = child(x, "y"); trans[2] = child(x, "z"); }
            if (const XmlNode* x = xo.first("rotation")) {
                rot[0] = child(x, "x"); rot[1] = child(x, "y"); rot[2] = child(x, "z"); angle = child(x, "angle");
            }
            if (const XmlNode* x = xo.first("scale")) { scl[0] = child(x, "x"); scl[1] = child(x, "y"); scl[2] = child(x, "z"); }
            M4 T = scale(rotate(translate(identity(), trans), angle, rot), scl);
            size_t first_face = s.m_vertices.size() / 3;
            std::vector<size_t> part_faces;
            if (!xf || !loadMesh(joinPath(base_dir, xf->text), s.m_vertices, part_faces, error)) {
                if (error.empty()) error = "Failed to load scene.";
                return false;
            }
            proto.kind = UBDPT_MODEL_MESH;
            std::memcpy(proto.W, T.c, sizeof(proto.W));
            for (size_t faces : part_faces) {  // one object (or emitter) per model of the file (:184-190)
                proto.first_face = first_face;
                proto.num_faces = faces;
                s.m_objects.push_back(proto);
                first_face += faces;
            }
        }
        // any other type: the reference adds nothing (:119-191)
        return true;
    }

    static std::shared_ptr<LoadedScene> build(const XmlNode& doc, const std::string& base_dir, std::string& error) {
        const XmlNode& root = *doc.children[0];
        auto s = std::make_shared<LoadedScene>();
        // camera (:25-55)
        const XmlNode* xc = root.first("camera");
        if (!xc) { error = "scene has no camera"; return nullptr; }
        const XmlNode* xr = xc->first("ratio");
        ubdpt_camera cam{};
        cam.ratio = child(xr, "w") / child(xr, "h");
        cam.vfov = child(xc, "vfov");
        cam.look_at[2] = 1.0;  // defaults: position (0,0,0), lookAt (0,0,1)
        if (const XmlNode* xp = xc->first("position")) { cam.pos[0] = child(xp, "x"); cam.pos[1] = child(xp, "y"); cam.pos[2] = child(xp, "z"); }
        if (const XmlNode* xl = xc->first("lookAt")) { cam.look_at[0] = child(xl, "x"); cam.look_at[1] = child(xl, "y"); cam.look_at[2] = child(xl, "z"); }
        s->m_desc.camera = cam;
        std::map<std::string, uint32_t> image_cache;
        for (const XmlNode* xo : root.byTag("object"))
            if (!object(*s, *xo, base_dir, image_cache, error)) return nullptr;
        s->finalize();
        return s;
    }
};

void LoadedScene::finalize() {
    m_desc.num_objects = (uint32_t)m_objects.size();
    m_desc.num_textures = (uint32_t)m_textures.size();
    m_desc.objects = m_objects.data();
    m_desc.textures = m_textures.data();
    m_desc.num_faces = m_vertices.size() / 3;
    m_desc.vertices = m_vertices.data();
    m_desc.num_texel_bytes = m_texels.size();
    m_desc.texels = m_texels.data();
}

bool LoadedScene::flatten(ubdpt_scene_desc& out) {
    out = m_desc;
    return true;
}

std::shared_ptr<LoadedScene> LoadedScene::fromXmlText(const std::string& xml_text, const std::string& base_dir, std::string& error) {
    XmlParser p(xml_text);
    auto doc = p.parse(error);
    if (!doc) { error = "Failed to load xml document. (" + error + ")"; return nullptr; }
    return SceneBuilderAccess::build(*doc, base_dir, error);
}

std::shared_ptr<LoadedScene> LoadedScene::fromXml(const std::string& xml_path, const std::string& base_dir, std::string& error) {
    std::ifstream in(xml_path);
    if (!in.is_open()) { error = "Failed to open scene file."; return nullptr; }
    std::stringstream ss;
    ss << in.rdbuf();
    return fromXmlText(ss.str(), base_dir, error);
}

}  // namespace uhip

// ---- C shim for the Python tests -------------------------------------------------------------------------
extern "C" {

struct uhip_scene {
    std::shared_ptr<uhip::LoadedScene> scene;
    std::string error;
};

uhip_scene* uhip_scene_load_xml(const char* xml_path, const char* base_dir) {
    auto* h = new (std::nothrow) uhip_scene();
    if (!h) return nullptr;
    try {  // no exception crosses the C boundary: a scene too large for the host's memory is a scene that fails to load
        h->scene = uhip::LoadedScene::fromXml(xml_path ? xml_path : "", base_dir ? base_dir : "", h->error);
    } catch (const std::exception& e) {
        h->scene.reset();
        h->error = std::string("loading failed: ") + e.what();
    }
    return h;
}
int uhip_scene_ok(const uhip_scene* h) { return h && h->scene ? 1 : 0; }
const char* uhip_scene_error(const uhip_scene* h) { return h ? h->error.c_str() : "null"; }
const ubdpt_scene_desc* uhip_scene_desc(const uhip_scene* h) { return h && h->scene ? &h->scene->desc() : nullptr; }
void uhip_scene_free(uhip_scene* h) { delete h; }

}  // extern "C"
