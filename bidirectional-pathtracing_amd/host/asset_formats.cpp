// asset_formats.cpp — see asset_formats.h.
#include "asset_formats.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>

namespace uhip {
namespace formats {

// largest image a reader accepts (2^28 pixels = 1 GiB of RGBA8): header fields beyond it are taken for damage
constexpr uint64_t kMaxImagePixels = 1ull << 28;

namespace {

bool readFile(const std::string& path, std::vector<uint8_t>& bytes) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) return false;
    in.seekg(0, std::ios::end);
    std::streamoff n = in.tellg();
    if (n < 0) return false;
    in.seekg(0, std::ios::beg);
    bytes.resize((size_t)n);
    if (n) in.read(reinterpret_cast<char*>(bytes.data()), n);
    return (std::streamoff)in.gcount() == n;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
uint32_t le16(const uint8_t* p) { return ((uint32_t)p[1] << 8) | p[0]; }

// ---- PLY ---------------------------------------------------------------------------------------------------------
enum PlyType { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
PlyType plyType(const std::string& s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
size_t plySize(PlyType t) {
    switch (t) {
        case T_I8: case T_U8: return 1;
        case T_I16: case T_U16: return 2;
        case T_I32: case T_U32: case T_F32: return 4;
        case T_F64: return 8;
        default: return 0;
    }
}
struct PlyProperty { std::string name; bool list = false; PlyType count_type = T_BAD, type = T_BAD; };
struct PlyElement { std::string name; size_t count = 0; std::vector<PlyProperty> props; };

// One value of a PLY body: text files give numbers separated by white space, binary files fixed-size fields.
struct PlyBody {
    const uint8_t* p;
    const uint8_t* end;
    int format;  // 0 text, 1 little endian, 2 big endian
    bool ok = true;
    double next(PlyType t) {
        if (format == 0) {
            while (p < end && isspace(*p)) p++;
            if (p >= end) { ok = false; return 0; }
            char* e = nullptr;
            std::string tok;
            const uint8_t* q = p;
            while (q < end && !isspace(*q)) q++;
            tok.assign(reinterpret_cast<const char*>(p), (size_t)(q - p));
            double v = std::strtod(tok.c_str(), &e);
            if (e == tok.c_str()) ok = false;
            p = q;
            return v;
        }
        const size_t n = plySize(t);
        if (n == 0 || (size_t)(end - p) < n) { ok = false; return 0; }
        uint8_t b[8];
        for (size_t k = 0; k < n; k++) b[k] = format == 1 ? p[k] : p[n - 1 - k];  // to little endian
        p += n;
        switch (t) {
            case T_I8: return (double)(int8_t)b[0];
            case T_U8: return (double)b[0];
            case T_I16: { int16_t v; std::memcpy(&v, b, 2); return (double)v; }
            case T_U16: { uint16_t v; std::memcpy(&v, b, 2); return (double)v; }
            case T_I32: { int32_t v; std::memcpy(&v, b, 4); return (double)v; }
            case T_U32: { uint32_t v; std::memcpy(&v, b, 4); return (double)v; }
            case T_F32: { float v; std::memcpy(&v, b, 4); return (double)v; }
            default: { double v; std::memcpy(&v, b, 8); return v; }
        }
    }
};

}  // namespace

std::string lowerExtension(const std::string& path) {
    size_t dot = path.find_last_of('.');
    size_t slash = path.find_last_of('/');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return std::string();
    std::string e = path.substr(dot + 1);
    for (char& c : e) c = (char)tolower((unsigned char)c);
    return e;
}

bool parsePly(const std::string& path, MeshSoup& out, std::string& error) {
    std::vector<uint8_t> bytes;
    if (!readFile(path, bytes)) { error = "cannot open mesh file " + path; return false; }
    // header: lines up to "end_header"
    size_t pos = 0;
    auto nextLine = [&](std::string& line) {
        if (pos >= bytes.size()) return false;
        size_t e = pos;
        while (e < bytes.size() && bytes[e] != '\n') e++;
        line.assign(reinterpret_cast<const char*>(bytes.data() + pos), e - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e < bytes.size() ? e + 1 : e;
        return true;
    };
    std::string line;
    if (!nextLine(line) || line != "ply") { error = "not a PLY file: " + path; return false; }
    int format = -1;
    std::vector<PlyElement> elements;
    bool header_done = false;
    while (nextLine(line)) {
        std::istringstream ls(line);
        std::string kw;
        ls >> kw;
        if (kw == "format") {
            std::string f;
            ls >> f;
            format = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
        } else if (kw == "element") {
            PlyElement e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (kw == "property" && !elements.empty()) {
            PlyProperty p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.list = true; p.count_type = plyType(ct); p.type = plyType(it);
                if (p.count_type == T_BAD) { error = "unknown property type in " + path; return false; }
            } else {
                p.type = plyType(t);
                ls >> p.name;
            }
            if (p.type == T_BAD) { error = "unknown property type in " + path; return false; }
            elements.back().props.push_back(p);
        } else if (kw == "end_header") { header_done = true; break; }
    }
    if (!header_done || format < 0) { error = "damaged PLY header in " + path; return false; }
    PlyBody body{bytes.data() + pos, bytes.data() + bytes.size(), format};
    std::vector<Corner> corners;
    bool have_n = false, have_uv = false;
    for (const PlyElement& el : elements) {
        if (el.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t k = 0; k < el.props.size(); k++) {
                const std::string& n = el.props[k].name;
                if (el.props[k].list) continue;
                if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
                else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
                else if (n == "s" || n == "u" || n == "texture_u") iu = (int)k;
                else if (n == "t" || n == "v" || n == "texture_v") iv = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) { error = "PLY vertices without x, y, z in " + path; return false; }
            have_n = inx >= 0 && iny >= 0 && inz >= 0;
            have_uv = iu >= 0 && iv >= 0;
            std::vector<double> val(el.props.size());
            for (size_t i = 0; i < el.count; i++) {
                for (size_t k = 0; k < el.props.size(); k++) {
                    const PlyProperty& pr = el.props[k];
                    if (pr.list) {
                        const double cnt = body.next(pr.count_type);
                        for (long j = 0; j < (long)cnt && body.ok; j++) body.next(pr.type);
                    } else
                        val[k] = body.next(pr.type);
                }
                if (!body.ok) { error = "PLY file ends inside its vertices: " + path; return false; }
                out.pos.push_back(V3{val[ix], val[iy], -val[iz]});  // left-handed: z negated
                if (have_n) out.nrm.push_back(V3{val[inx], val[iny], -val[inz]});
                if (have_uv) out.uv.emplace_back(val[iu], val[iv]);
            }
        } else if (el.name == "face") {
            for (size_t i = 0; i < el.count; i++) {
                for (const PlyProperty& pr : el.props) {
                    if (!pr.list) { body.next(pr.type); continue; }
                    const long cnt = (long)body.next(pr.count_type);
                    if (!body.ok || cnt < 0 || cnt > 1 << 20) { error = "damaged PLY face list in " + path; return false; }
                    std::vector<long> idx((size_t)cnt);
                    for (long j = 0; j < cnt; j++) idx[(size_t)j] = (long)body.next(pr.type);
                    if (pr.name != "vertex_indices" && pr.name != "vertex_index") continue;
                    for (long j = 1; j + 1 < cnt; j++)  // aiProcess_Triangulate: fan
                        for (long k : {0L, j, j + 1}) {
                            const long v = idx[(size_t)k];
                            if (v < 0 || v >= (long)out.pos.size()) { error = "face index out of range in " + path; return false; }
                            corners.push_back(Corner{(int)v + 1, have_uv ? (int)v + 1 : 0, have_n ? (int)v + 1 : 0});
                        }
                }
                if (!body.ok) { error = "PLY file ends inside its faces: " + path; return false; }
            }
        } else {  // an element this loader has no use for: skipped field by field
            for (size_t i = 0; i < el.count && body.ok; i++)
                for (const PlyProperty& pr : el.props) {
                    if (!pr.list) { body.next(pr.type); continue; }
                    const long cnt = (long)body.next(pr.count_type);
                    for (long j = 0; j < cnt && body.ok; j++) body.next(pr.type);
                }
        }
    }
    if (corners.empty()) { error = "mesh has no faces: " + path; return false; }
    out.parts.push_back(std::move(corners));
    return true;
}

// STL: a list of facets, each with a normal and three corner positions; no shared vertices, no texture coordinates.
// A facet's normal becomes the normal of its three corners (a facet whose normal is the zero vector gets the one its
// corners span). Binary files are told from text files by their length: 84 bytes + 50 per facet.
bool parseStl(const std::string& path, MeshSoup& out, std::string& error) {
    std::vector<uint8_t> bytes;
    if (!readFile(path, bytes)) { error = "cannot open mesh file " + path; return false; }
    std::vector<Corner> corners;
    auto facet = [&](V3 n, const V3 (&p)[3]) {
        V3 q[3];
        for (int k = 0; k < 3; k++) q[k] = V3{p[k].x, p[k].y, -p[k].z};  // left-handed: z negated
        n.z = -n.z;
        if (n.x == 0 && n.y == 0 && n.z == 0) {
            const V3 a{q[1].x - q[0].x, q[1].y - q[0].y, q[1].z - q[0].z}, b{q[2].x - q[0].x, q[2].y - q[0].y, q[2].z - q[0].z};
            n = V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
        }
        out.nrm.push_back(n);
        for (int k = 0; k < 3; k++) {
            out.pos.push_back(q[k]);
            corners.push_back(Corner{(int)out.pos.size(), 0, (int)out.nrm.size()});
        }
    };
    const bool binary = bytes.size() >= 84 && bytes.size() == 84 + 50 * (size_t)le32(bytes.data() + 80);
    if (binary) {
        const size_t n = le32(bytes.data() + 80);
        for (size_t f = 0; f < n; f++) {
            float v[12];
            std::memcpy(v, bytes.data() + 84 + 50 * f, sizeof(v));
            const V3 p[3] = {{v[3], v[4], v[5]}, {v[6], v[7], v[8]}, {v[9], v[10], v[11]}};
            facet(V3{v[0], v[1], v[2]}, p);
        }
    } else {
        std::istringstream in(std::string(reinterpret_cast<const char*>(bytes.data()), bytes.size()));
        std::string tok;
        if (!(in >> tok) || tok != "solid") { error = "not an STL file: " + path; return false; }
        V3 n{0, 0, 0}, p[3];
        int have = 0;
        while (in >> tok) {
            if (tok == "facet") {
                std::string kw;
                in >> kw >> n.x >> n.y >> n.z;
                have = 0;
            } else if (tok == "vertex") {
                V3 v{0, 0, 0};
                in >> v.x >> v.y >> v.z;
                if (have < 3) p[have] = v;
                have++;
            } else if (tok == "endfacet") {
                if (have != 3 || in.fail()) { error = "damaged STL facet in " + path; return false; }
                facet(n, p);
            }
        }
    }
    if (corners.empty()) { error = "mesh has no faces: " + path; return false; }
    out.parts.push_back(std::move(corners));
    return true;
}

// ---- images ------------------------------------------------------------------------------------------------------
namespace {

// 16-bit sample -> 8 bits with rounding (x / 257)
uint8_t from16(uint32_t x) { return (uint8_t)((x * 255u + 32767u) / 65535u); }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undoes the per-row filters of one PNG (sub-)image of `rows` rows of `row_bytes` bytes in place; `src` holds a
// filter byte before every row. bpp = bytes per complete pixel, at least 1.
bool pngUnfilter(const uint8_t* src, size_t rows, size_t row_bytes, size_t bpp, std::vector<uint8_t>& dst) {
    dst.assign(rows * row_bytes, 0);
    for (size_t y = 0; y < rows; y++) {
        const uint8_t ft = src[y * (row_bytes + 1)];
        const uint8_t* in = src + y * (row_bytes + 1) + 1;
        uint8_t* cur = dst.data() + y * row_bytes;
        const uint8_t* up = y ? cur - row_bytes : nullptr;
        if (ft > 4) return false;
        for (size_t x = 0; x < row_bytes; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            if (ft == 1) v += a;
            else if (ft == 2) v += b;
            else if (ft == 3) v += (a + b) / 2;
            else if (ft == 4) v += paeth(a, b, c);
            cur[x] = (uint8_t)v;
        }
    }
    return true;
}

bool loadPng(const std::vector<uint8_t>& f, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0) return false;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_ihdr = false;
    for (size_t p = 8; p + 12 <= f.size();) {
        const uint32_t len = be32(f.data() + p);
        if (len > f.size() - p - 12) return false;
        const uint8_t* type = f.data() + p + 4;
        const uint8_t* data = f.data() + p + 8;
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            W = be32(data); H = be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) return false;
            seen_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!seen_ihdr || W == 0 || H == 0 || W > (1u << 16) || H > (1u << 16) || (uint64_t)W * H > kMaxImagePixels) return false;
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return false;
    }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
    if (!depth_ok || (ctype == 3 && plte.size() < 3)) return false;
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = std::max<size_t>(1, bits_pp / 8);
    auto rowBytes = [&](size_t pixels) { return (pixels * bits_pp + 7) / 8; };
    // the sub-images of the file: one for a plain file, seven for an interlaced one
    struct Pass { size_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass* passes = interlace ? adam7 : whole;
    const int npass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < npass; k++) {
        const size_t pw = (W > passes[k].x0) ? (W - passes[k].x0 + passes[k].dx - 1) / passes[k].dx : 0;
        const size_t ph = (H > passes[k].y0) ? (H - passes[k].y0 + passes[k].dy - 1) / passes[k].dy : 0;
        if (pw && ph) raw_size += ph * (rowBytes(pw) + 1);
    }
    // deflate expands by at most 1032 : 1, so a header that promises more than the compressed data can hold is damaged
    if (idat.empty() || raw_size / 1032 > idat.size()) return false;
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    if (idat.empty() || uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size) return false;
    w = W; h = H;
    rgba.assign((size_t)W * H * 4, 255);
    size_t off = 0;
    std::vector<uint8_t> img;
    for (int k = 0; k < npass; k++) {
        const Pass& ps = passes[k];
        const size_t pw = (W > ps.x0) ? (W - ps.x0 + ps.dx - 1) / ps.dx : 0;
        const size_t ph = (H > ps.y0) ? (H - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (!pw || !ph) continue;
        const size_t rb = rowBytes(pw);
        if (!pngUnfilter(raw.data() + off, ph, rb, bpp, img)) return false;
        off += ph * (rb + 1);
        for (size_t y = 0; y < ph; y++) {
            const uint8_t* row = img.data() + y * rb;
            for (size_t x = 0; x < pw; x++) {
                uint32_t s[4] = {0, 0, 0, 0};  // the pixel's samples, still at the file's bit depth
                for (int c = 0; c < channels; c++) {
                    if (depth == 8) s[c] = row[x * channels + c];
                    else if (depth == 16) s[c] = ((uint32_t)row[2 * (x * channels + c)] << 8) | row[2 * (x * channels + c) + 1];
                    else {
                        const size_t bit = x * (size_t)depth;
                        s[c] = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u);
                    }
                }
                uint8_t* o = rgba.data() + 4 * ((ps.y0 + y * ps.dy) * (size_t)W + ps.x0 + x * ps.dx);
                auto to8 = [&](uint32_t v) -> uint8_t {
                    if (depth == 16) return from16(v);
                    if (depth == 8) return (uint8_t)v;
                    return (uint8_t)(v * 255u / ((1u << depth) - 1u));
                };
                if (ctype == 3) {
                    if (3 * (size_t)s[0] + 2 >= plte.size()) return false;
                    o[0] = plte[3 * s[0]]; o[1] = plte[3 * s[0] + 1]; o[2] = plte[3 * s[0] + 2];
                } else if (channels <= 2) {
                    o[0] = o[1] = o[2] = to8(s[0]);
                    if (channels == 2) o[3] = to8(s[1]);
                } else {
                    o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
                    if (channels == 4) o[3] = to8(s[3]);
                }
            }
        }
    }
    return true;
}

bool loadBmp(const std::vector<uint8_t>& f, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    if (f.size() < 14 + 40 || f[0] != 'B' || f[1] != 'M') return false;
    const size_t data_off = le32(f.data() + 10);
    const uint32_t hdr = le32(f.data() + 14);
    if (hdr < 40 || 14 + (size_t)hdr > f.size()) return false;
    const int32_t W = (int32_t)le32(f.data() + 18);
    const int32_t Hs = (int32_t)le32(f.data() + 22);
    const uint32_t bpp = le16(f.data() + 28), comp = le32(f.data() + 30);
    uint32_t ncol = le32(f.data() + 46);
    const bool top_down = Hs < 0;
    const int64_t H = top_down ? -(int64_t)Hs : Hs;
    if (W <= 0 || H <= 0 || W > (1 << 16) || H > (1 << 16) || (uint64_t)W * (uint64_t)H > kMaxImagePixels) return false;
    if (!(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32)) return false;
    if (!(comp == 0 || (comp == 3 && bpp == 32))) return false;
    uint32_t mask[3] = {0x00ff0000u, 0x0000ff00u, 0x000000ffu};
    if (comp == 3) {
        const size_t mo = hdr >= 52 ? 14 + 40 : 14 + (size_t)hdr;  // inside a V4/V5 header, else right behind a 40-byte one
        if (mo + 12 > f.size()) return false;
        for (int k = 0; k < 3; k++) mask[k] = le32(f.data() + mo + 4 * k);
    }
    const uint8_t* pal = nullptr;
    if (bpp <= 8) {
        if (ncol == 0) ncol = 1u << bpp;
        const size_t po = 14 + (size_t)hdr;
        if (po + 4 * (size_t)ncol > f.size()) return false;
        pal = f.data() + po;
    }
    const size_t stride = (((size_t)W * bpp + 31) / 32) * 4;
    if (data_off > f.size() || stride * (size_t)H > f.size() - data_off) return false;
    auto field = [](uint32_t v, uint32_t m) -> uint8_t {  // the masked bits scaled to 8
        if (!m) return 0;
        int sh = 0;
        while (!((m >> sh) & 1u)) sh++;
        const uint32_t top = m >> sh;
        return (uint8_t)(((v & m) >> sh) * 255u / top);
    };
    w = (uint32_t)W; h = (uint32_t)H;
    rgba.assign((size_t)W * H * 4, 255);
    for (int64_t y = 0; y < H; y++) {
        const uint8_t* row = f.data() + data_off + stride * (size_t)(top_down ? y : H - 1 - y);
        for (int32_t x = 0; x < W; x++) {
            uint8_t* o = rgba.data() + 4 * ((size_t)y * W + x);
            if (bpp == 24) { o[0] = row[3 * x + 2]; o[1] = row[3 * x + 1]; o[2] = row[3 * x]; }
            else if (bpp == 32) {
                const uint32_t v = le32(row + 4 * (size_t)x);
                for (int k = 0; k < 3; k++) o[k] = field(v, mask[k]);
            } else {
                const size_t bit = (size_t)x * bpp;
                const uint32_t idx = (row[bit / 8] >> (8 - bpp - (bit % 8))) & ((1u << bpp) - 1u);
                if (idx >= ncol) return false;
                o[0] = pal[4 * idx + 2]; o[1] = pal[4 * idx + 1]; o[2] = pal[4 * idx];
            }
        }
    }
    return true;
}

bool loadTga(const std::vector<uint8_t>& f, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    if (f.size() < 18) return false;
    const uint32_t idlen = f[0], cmap = f[1], type = f[2];
    const uint32_t W = le16(f.data() + 12), H = le16(f.data() + 14), bpp = f[16], desc = f[17];
    if (cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || W == 0 || H == 0) return false;
    const bool grey = type == 3 || type == 11, rle = type >= 10;
    if (!(grey ? bpp == 8 : (bpp == 24 || bpp == 32))) return false;
    const size_t px = bpp / 8;
    size_t p = 18 + (size_t)idlen, o = 0;
    // before anything is sized from the header: the pixel data must be able to come out of what the file holds (a
    // run-length packet of 1 + px bytes makes at most 128 pixels)
    if (p >= f.size() || (uint64_t)W * H > kMaxImagePixels) return false;
    const uint64_t avail = f.size() - p;
    if (rle ? ((uint64_t)W * H + 127) / 128 * (1 + px) > avail + px : (uint64_t)W * H * px > avail) return false;
    std::vector<uint8_t> pix((size_t)W * H * px);
    if (!rle) {
        if (p + pix.size() > f.size()) return false;
        std::memcpy(pix.data(), f.data() + p, pix.size());
    } else {
        while (o < pix.size()) {
            if (p >= f.size()) return false;
            const uint32_t head = f[p++], n = (head & 127u) + 1;
            if (o + (size_t)n * px > pix.size()) return false;
            if (head & 128u) {
                if (p + px > f.size()) return false;
                for (uint32_t k = 0; k < n; k++, o += px) std::memcpy(pix.data() + o, f.data() + p, px);
                p += px;
            } else {
                if (p + (size_t)n * px > f.size()) return false;
                std::memcpy(pix.data() + o, f.data() + p, (size_t)n * px);
                p += (size_t)n * px; o += (size_t)n * px;
            }
        }
    }
    const bool top = (desc & 0x20u) != 0, right = (desc & 0x10u) != 0;
    w = W; h = H;
    rgba.assign((size_t)W * H * 4, 255);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            const uint8_t* s = pix.data() + px * ((size_t)(top ? y : H - 1 - y) * W + (right ? W - 1 - x : x));
            uint8_t* d = rgba.data() + 4 * ((size_t)y * W + x);
            if (grey) d[0] = d[1] = d[2] = s[0];
            else { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; if (px == 4) d[3] = s[3]; }
        }
    return true;
}

}  // namespace

bool loadImageFile(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    // a file that cannot be read is a texture that fails to load (as QImage::load does for the reference,
    // uapp/textureimg.cpp:7-17), never an exception that leaves through the C entry points
    try {
        std::vector<uint8_t> f;
        if (!readFile(path, f) || f.size() < 4) return false;
        if (f[0] == 0x89 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G') return loadPng(f, w, h, rgba);
        if (f[0] == 'B' && f[1] == 'M') return loadBmp(f, w, h, rgba);
        if (lowerExtension(path) == "tga") return loadTga(f, w, h, rgba);
        return false;
    } catch (const std::bad_alloc&) {
        rgba.clear();
        return false;
    }
}

}  // namespace formats
}  // namespace uhip
