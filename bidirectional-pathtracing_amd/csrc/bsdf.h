// bsdf.h — random numbers, sampling maps, textures, material lobes and BSDFs on the device.
//
// BSDF routines follow UBsdf::samplePSA / pPSA / scatter (uengine/ubsdf.h:22-38 and
// ubsdflambertian.cpp, ubsdfperfectmirror.cpp, ubsdfdielectric.cpp): tangent space with y along the
// shading normal, densities in projected-solid-angle measure.
#pragma once

#include "trace.h"

namespace ub {

// ---- counter-based uniforms ------------------------------------------------------------------------
// The reference draws from one shared mt19937 (uengine/umath.cpp:10-20). A GPU needs independent,
// order-free streams: u = hash(seed, pass, pixel, stream, dimension). Only the distribution is
// contractual: [0,1) with 53 random bits, like uniform_real_distribution<double>(0,1).
DEV uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
DEV uint64_t pathKey(uint64_t seed, uint64_t pass, uint64_t pixel) {
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (pass + 1));
    return mix64(k ^ (pixel * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}
struct Rng {
    uint64_t key;
    uint32_t stream;
    DEV double operator()(uint32_t dim) const {
        uint64_t ctr = (((uint64_t)stream << 32) | dim) + 1;
        uint64_t h = mix64(key + 0x9E3779B97F4A7C15ULL * ctr);
        return (double)(h >> 11) * 0x1.0p-53;
    }
};
enum { STREAM_EYE = 0, STREAM_LIGHT = 1 };
enum { DIM_LENS_X = 0, DIM_LENS_Y = 1, DIM_PIXEL_X = 2, DIM_PIXEL_Y = 3 };
enum { DIM_EMITTER = 0, DIM_POINT_A = 1, DIM_POINT_B = 2, DIM_POINT_C = 3, DIM_EMIT_ANGLE = 4, DIM_EMIT_V = 5 };
// the h-th intersection of a subpath (h = 1, 2, ...) owns dimensions 8h .. 8h+3
enum { DIM_LOBE = 0, DIM_SCATTER_1 = 1, DIM_SCATTER_2 = 2, DIM_ROULETTE = 3 };
DEV uint32_t hitDim(uint32_t h, uint32_t which) { return 8u * h + which; }

// ---- URng sampling maps (uengine/umath.cpp:22-86) ----------------------------------------------------
constexpr double kPi = 3.14159265358979323846;  // M_PI

DEV d2 sampleUnitRectStratified(uint32_t num_strata, uint32_t stratum_id, double u1, double u2) {
    uint32_t num_divs = (uint32_t)sqrt((double)num_strata);
    double d = 1.0 / (double)num_divs;
    uint32_t x = stratum_id % num_divs;
    uint32_t y = stratum_id / num_divs;
    d2 r;
    r.x = d * ((double)x + u1);
    r.y = d * ((double)y + u2);
    return r;
}
DEV d2 sampleUnitDiskStratified(uint32_t num_strata, uint32_t stratum_id, double u1, double u2) {
    d2 uv = sampleUnitRectStratified(num_strata, stratum_id, u1, u2);
    double theta = 2 * kPi * uv.x;
    double r = sqrt(uv.y);
    d2 o;
    o.x = r * cos(theta);
    o.y = r * sin(theta);
    return o;
}
DEV d3 samplePosHemUniform(double u1, double u2) {
    double angle = u1 * kPi * 2.0;
    double v = u2;
    double s = sqrt(1 - v * v);
    return mk3(cos(angle) * s, v, sin(angle) * s);
}
DEV d3 samplePosHemCos(double u1, double u2) {
    double angle = u1 * kPi * 2.0;
    double s = u2;
    double y = sqrt(s);
    double r = sqrt(1 - s);
    return mk3(r * cos(angle), y, r * sin(angle));
}
DEV d3 sampleUnitSphereUniform(double u1, double u2) {
    double u = u1 * 2.0 * kPi;
    double v = u2 * 2.0 - 1.0;
    double r = sqrt(1 - (v * v));
    return mk3(cos(u) * r, v, sin(u) * r);
}
DEV d3 sampleTriangleUniform(d3 p0, d3 p1, d3 p2, double r1, double r2, double& u, double& v) {
    double s = sqrt(r1);
    double m = s * r2;
    u = s - m;
    v = m;
    return (1.0 - s) * p0 + (s - m) * p1 + m * p2;
}

// ---- textures (uapp/texturecolor.cpp:8-11, uapp/textureimg.cpp:56-82) --------------------------------
DEV d3 texel(const uint8_t* px) {  // uapp/textureimg.cpp:28-32
    return mk3((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
}
DEV d3 textureSample(const DScene& S, uint32_t t, double u, double v) {
    const DTexture& tx = S.textures[t];
    if (tx.kind == 0) return tx.color;
    // the reference subtracts size_t(u) from BOTH coordinates (textureimg.cpp:59-60)
    u = u - (double)(unsigned long long)u;
    v = v - (double)(unsigned long long)u;
    double x = u * (double)(tx.width - 1);
    double y = v * (double)(tx.height - 1);
    unsigned long long x1 = (unsigned long long)floor(x), x2 = (unsigned long long)ceil(x);
    unsigned long long y1 = (unsigned long long)floor(y), y2 = (unsigned long long)ceil(y);
    double tx_ = x - (double)x1;
    double ty_ = y - (double)y1;
    // indices the reference would assert on (textureimg.cpp:73-76) are clamped instead of faulting
    unsigned long long wm = tx.width - 1, hm = tx.height - 1;
    x1 = x1 > wm ? wm : x1; x2 = x2 > wm ? wm : x2;
    y1 = y1 > hm ? hm : y1; y2 = y2 > hm ? hm : y2;
    const uint8_t* base = S.texels + tx.texel_offset;
    unsigned long long w = tx.width;
    // the reference's weights are swapped: tx multiplies the x1 texel (textureimg.cpp:78-81)
    d3 c1 = tx_ * texel(base + 4 * (w * y1 + x1)) + (1.0 - tx_) * texel(base + 4 * (w * y1 + x2));
    d3 c2 = tx_ * texel(base + 4 * (w * y2 + x1)) + (1.0 - tx_) * texel(base + 4 * (w * y2 + x2));
    return ty_ * c1 + (1.0 - ty_) * c2;
}

// ---- material -> lobe (uapp/material.h:25-31,45-48,66-79,97-100) -------------------------------------
enum { LOBE_NONE = -1, LOBE_LAMBERTIAN = 0, LOBE_MIRROR = 1, LOBE_DIELECTRIC = 2 };
DEV int pickLobe(const DObject& o, double r) {
    switch (o.material) {
    case 0: return (r < 0.8) ? LOBE_LAMBERTIAN : LOBE_NONE;  // LatexPaint: 20 % absorbed, not compensated
    case 1:                                                   // Glossy(d, s)
        if (r < o.mat_a) return LOBE_LAMBERTIAN;
        r -= o.mat_a;
        if (r < o.mat_b) return LOBE_MIRROR;
        return LOBE_NONE;
    case 2: return LOBE_MIRROR;
    case 3: return LOBE_DIELECTRIC;
    }
    return LOBE_NONE;
}

// ---- BSDFs --------------------------------------------------------------------------------------------
struct Frame {  // UBsdfSurfaceInfo (ubsdf.h:8-20); eta_t is always 1 (ubsdf.cpp:13)
    d3 Ng, Ns, Ts, Bs;
};
constexpr double kEtaT = 1.0;

// ubsdfdielectric.cpp:3-9
DEV double fresnel(double eta_i, double eta_t, double cos_i, double cos_t) {
    double rp = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    double rs = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    return 0.5 * (rp * rp + rs * rs);
}

// UBsdf::samplePSA
DEV d3 bsdfEval(int lobe, double eta, d3 tex, const Frame& fr, d3 wiT, d3 woT) {
    if (lobe == LOBE_LAMBERTIAN || lobe == LOBE_MIRROR) {  // ubsdflambertian.cpp:10-26, ubsdfperfectmirror.cpp:8-24
        m3 TNB = mk_m3(fr.Ts, fr.Ns, fr.Bs);
        d3 wiL = normalize(mul(TNB, wiT));
        d3 woL = normalize(mul(TNB, woT));
        if (dot(fr.Ng, wiL) * dot(fr.Ng, woL) <= 0) return mk3(0, 0, 0);
        if (wiT.y * woT.y <= 0) return mk3(0, 0, 0);
        return lobe == LOBE_LAMBERTIAN ? (1.0 / kPi) * tex : tex;
    }
    if (lobe == LOBE_DIELECTRIC) {  // ubsdfdielectric.cpp:17-62
        double n, nt, Ny;
        if (wiT.y < 0) { Ny = -1.0; n = eta; nt = kEtaT; }
        else           { Ny = 1.0;  n = kEtaT; nt = eta; }
        double c = ((n * n) / (nt * nt)) * (1.0 - wiT.y * wiT.y);
        if (c > 1.0) return (wiT.y * woT.y <= 0) ? mk3(0, 0, 0) : tex;
        double R = fresnel(n, nt, Ny * wiT.y, sqrt(1.0 - c));
        double T = 1.0 - R;
        return (wiT.y * woT.y <= 0) ? T * tex : R * tex;
    }
    return mk3(0, 0, 0);
}

// UBsdf::pPSA
DEV double bsdfPdf(int lobe, double eta, const Frame& fr, d3 wsT, d3 wgT) {
    if (lobe == LOBE_LAMBERTIAN || lobe == LOBE_MIRROR) {  // ubsdflambertian.cpp:28-49, ubsdfperfectmirror.cpp:26-42
        m3 TNB = mk_m3(fr.Ts, fr.Ns, fr.Bs);
        d3 wsL = normalize(mul(TNB, wsT));
        d3 wgL = normalize(mul(TNB, wgT));
        if (dot(fr.Ng, wsL) * dot(fr.Ng, wgL) <= 0) return 0;
        if (wsT.y * wgT.y <= 0) return 0;
        return lobe == LOBE_LAMBERTIAN ? (1.0 / kPi) : 1.0;  // cosine-weighted Lambertian (material.h:22,60)
    }
    if (lobe == LOBE_DIELECTRIC) {  // ubsdfdielectric.cpp:64-108
        double n, nt, Ny;
        if (wgT.y < 0) { Ny = -1.0; n = eta; nt = kEtaT; }
        else           { Ny = 1.0;  n = kEtaT; nt = eta; }
        double c = ((n * n) / (nt * nt)) * (1.0 - wgT.y * wgT.y);
        if (c > 1.0) return (wsT.y * wgT.y <= 0) ? 0.0 : 1.0;
        double R = fresnel(n, nt, Ny * wgT.y, sqrt(1.0 - c));
        double T = 1.0 - R;
        return (wsT.y * wgT.y <= 0) ? T : R;
    }
    return 0;
}

// UBsdf::scatter; u1, u2 are the uniforms the reference would draw, in its order
DEV bool bsdfScatter(int lobe, double eta, d3 tex, const Frame& fr, d3 w, double u1, double u2, d3& dirT,
                     double& pdf, d3& fs, bool& specular) {
    if (lobe < 0) return false;
    if (dot(w, fr.Ns) * dot(w, fr.Ng) <= 0) return false;
    d3 wT = normalize(rowmul(w, mk_m3(fr.Ts, fr.Ns, fr.Bs)));
    if (lobe == LOBE_LAMBERTIAN) {  // ubsdflambertian.cpp:51-81
        dirT = samplePosHemCos(u1, u2);
        pdf = 1.0 / kPi;
        if (wT.y < 0) dirT = dirT * -1.0;
        specular = false;
        fs = (1.0 / kPi) * tex;
        return true;
    }
    if (lobe == LOBE_MIRROR) {  // ubsdfperfectmirror.cpp:44-66
        dirT = normalize(reflect(-wT, mk3(0, 1, 0)));
        if (wT.y < 0) dirT = dirT * -1.0;  // the reference negates the reflected direction below the surface
        specular = true;
        pdf = 1.0;
        fs = tex;
        return true;
    }
    // ubsdfdielectric.cpp:110-174
    specular = true;
    double n, nt;
    d3 N;
    if (wT.y < 0) { N = mk3(0.0, -1.0, 0.0); n = eta; nt = kEtaT; }
    else          { N = mk3(0.0, 1.0, 0.0);  n = kEtaT; nt = eta; }
    double eta_r = n / nt;
    double c = (eta_r * eta_r) * (1.0 - wT.y * wT.y);
    double R, T, c1 = 0;
    if (c > 1.0) R = 1.0;
    else { c1 = sqrt(1.0 - c); R = fresnel(n, nt, N.y * wT.y, c1); }
    T = 1.0 - R;
    if (u1 < R) {
        dirT = normalize(reflect(-wT, N));
        pdf = R;
        fs = R * tex;
    } else {
        dirT = normalize(eta_r * (-wT) - N * (eta_r * dot(N, -wT) + c1));
        pdf = T;
        fs = T * tex;
    }
    return true;
}

}  // namespace ub
