// dmath.h — double-precision vector / matrix arithmetic for host and gfx950 device code.
//
// The reference does all of its arithmetic through GLM (uengine/umath.h:4-12). To keep hit
// identities bit-exact with it, every routine here evaluates in GLM's operation order and the
// translation units that include this header are compiled with -ffp-contract=off (hipcc would
// otherwise fuse a*b+c into v_fma_f64).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define HD __host__ __device__ __forceinline__

namespace ub {

struct d2 { double x, y; };
struct d3 { double x, y, z; };
struct d4 { double x, y, z, w; };
struct f3 { float x, y, z; };
struct m4 { d4 c[4]; };  // column-major: c[i] = column i (glm::dmat4x4 m[i])
struct m3 { d3 c[3]; };
struct fm3 { f3 c[3]; };

HD d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
HD d4 mk4(double x, double y, double z, double w) { d4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
HD d4 mk4(d3 v, double w) { return mk4(v.x, v.y, v.z, w); }
HD d3 xyz(d4 v) { return mk3(v.x, v.y, v.z); }

HD d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD d3 operator-(d3 a) { return mk3(-a.x, -a.y, -a.z); }
HD d3 operator*(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HD d3 operator*(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
HD d3 operator*(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
HD d3 operator/(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }

HD d4 operator+(d4 a, d4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
HD d4 operator-(d4 a, d4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
HD d4 operator*(d4 a, d4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
HD d4 operator*(d4 a, double s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }

// glm::dot: products summed left to right; vec4: (x+y)+(z+w)
HD double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD d3 cross(d3 x, d3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
HD double length(d3 a) { return sqrt(dot(a, a)); }
HD double distance(d3 p0, d3 p1) { return length(p1 - p0); }
// glm::normalize: v * (1 / sqrt(dot(v, v)))
HD d3 normalize(d3 a) { return a * (1.0 / sqrt(dot(a, a))); }
// glm::reflect: I - N * dot(N, I) * 2
HD d3 reflect(d3 I, d3 N) { return I - N * dot(N, I) * 2.0; }

// mat4 * vec4, xyz part only: (m0*x + m1*y) + (m2*z + m3*w)
HD d3 mul_xyz(const m4& m, d3 v, double w) {
    d3 a0 = xyz(m.c[0]) * v.x + xyz(m.c[1]) * v.y;
    d3 a1 = xyz(m.c[2]) * v.z + xyz(m.c[3]) * w;
    return a0 + a1;
}
HD d4 mul(const m4& m, d4 v) {
    d4 a0 = m.c[0] * v.x + m.c[1] * v.y;
    d4 a1 = m.c[2] * v.z + m.c[3] * v.w;
    return a0 + a1;
}
// vec4(v, 0) * mat4 (row vector), xyz part: component i = m[i].x*v.x + m[i].y*v.y + m[i].z*v.z + m[i].w*0
HD d3 rowmul_xyz0(d3 v, const m4& m) {
    return mk3(m.c[0].x * v.x + m.c[0].y * v.y + m.c[0].z * v.z + m.c[0].w * 0.0,
               m.c[1].x * v.x + m.c[1].y * v.y + m.c[1].z * v.z + m.c[1].w * 0.0,
               m.c[2].x * v.x + m.c[2].y * v.y + m.c[2].z * v.z + m.c[2].w * 0.0);
}
HD d3 mul(const m3& m, d3 v) {
    return mk3(m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
               m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
               m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z);
}
// vec3 * mat3 (row vector) = (dot(c0, v), dot(c1, v), dot(c2, v))
HD d3 rowmul(d3 v, const m3& m) {
    return mk3(m.c[0].x * v.x + m.c[0].y * v.y + m.c[0].z * v.z,
               m.c[1].x * v.x + m.c[1].y * v.y + m.c[1].z * v.z,
               m.c[2].x * v.x + m.c[2].y * v.y + m.c[2].z * v.z);
}
HD m3 mk_m3(d3 c0, d3 c1, d3 c2) { m3 r; r.c[0] = c0; r.c[1] = c1; r.c[2] = c2; return r; }

// The reference builds its tangent frame matrix as a FLOAT glm::mat3x3 in both subpath generators
// (uengine/ubdptrenderer.cpp:287,448) and multiplies the sampled direction in float.
HD d3 float_tnb_mul(d3 Ts, d3 Ns, d3 Bs, d3 v) {
    float c0x = (float)Ts.x, c0y = (float)Ts.y, c0z = (float)Ts.z;
    float c1x = (float)Ns.x, c1y = (float)Ns.y, c1z = (float)Ns.z;
    float c2x = (float)Bs.x, c2y = (float)Bs.y, c2z = (float)Bs.z;
    float vx = (float)v.x, vy = (float)v.y, vz = (float)v.z;
    float rx = c0x * vx + c1x * vy + c2x * vz;
    float ry = c0y * vx + c1y * vy + c2y * vz;
    float rz = c0z * vx + c1z * vy + c2z * vz;
    return mk3((double)rx, (double)ry, (double)rz);
}

// ugeometry.cpp:116-136
HD d3 transformPoint(const m4& T, d3 P) { return mul_xyz(T, P, 1.0); }
HD d3 transformVector(const m4& T, d3 V) { return normalize(mul_xyz(T, V, 0.0)); }
HD d3 transformVectorT(const m4& T, d3 V) { return normalize(rowmul_xyz0(V, T)); }

HD m4 identity4() {
    m4 r;
    r.c[0] = mk4(1, 0, 0, 0); r.c[1] = mk4(0, 1, 0, 0); r.c[2] = mk4(0, 0, 1, 0); r.c[3] = mk4(0, 0, 0, 1);
    return r;
}

// glm::inverse(dmat4): cofactors times reciprocal determinant (host side, scene flattening)
inline m4 inverse(const m4& mm) {
    const double(*m)[4] = reinterpret_cast<const double(*)[4]>(&mm);
    double c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    double c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    double c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    double c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    double c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    double c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    double c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    double c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    double c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    double c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    double c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    double c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    double c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    double c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    double c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    double c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    double c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    double c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    d4 F0 = mk4(c00, c00, c02, c03), F1 = mk4(c04, c04, c06, c07), F2 = mk4(c08, c08, c10, c11);
    d4 F3 = mk4(c12, c12, c14, c15), F4 = mk4(c16, c16, c18, c19), F5 = mk4(c20, c20, c22, c23);
    d4 V0 = mk4(m[1][0], m[0][0], m[0][0], m[0][0]);
    d4 V1 = mk4(m[1][1], m[0][1], m[0][1], m[0][1]);
    d4 V2 = mk4(m[1][2], m[0][2], m[0][2], m[0][2]);
    d4 V3 = mk4(m[1][3], m[0][3], m[0][3], m[0][3]);
    d4 I0 = V1 * F0 - V2 * F1 + V3 * F2;
    d4 I1 = V0 * F0 - V2 * F3 + V3 * F4;
    d4 I2 = V0 * F1 - V1 * F3 + V3 * F5;
    d4 I3 = V0 * F2 - V1 * F4 + V2 * F5;
    d4 SA = mk4(+1, -1, +1, -1), SB = mk4(-1, +1, -1, +1);
    m4 inv;
    inv.c[0] = I0 * SA; inv.c[1] = I1 * SB; inv.c[2] = I2 * SA; inv.c[3] = I3 * SB;
    d4 row0 = mk4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    d4 dot0 = mm.c[0] * row0;
    double det = (dot0.x + dot0.y) + (dot0.z + dot0.w);
    double ood = 1.0 / det;
    m4 r;
    for (int i = 0; i < 4; i++) r.c[i] = inv.c[i] * ood;
    return r;
}

// glm::translate / glm::scale (used for the mesh bounding-sphere matrix, uapp/mesh.cpp:85)
inline m4 translate(const m4& m, d3 v) {
    m4 r = m;
    r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
    return r;
}
inline m4 scale(const m4& m, d3 v) {
    m4 r;
    r.c[0] = m.c[0] * v.x; r.c[1] = m.c[1] * v.y; r.c[2] = m.c[2] * v.z; r.c[3] = m.c[3];
    return r;
}
// glm::lookAtLH (uengine/ucamera.cpp:13)
inline m4 lookAtLH(d3 eye, d3 center, d3 up) {
    d3 f = normalize(center - eye);
    d3 s = normalize(cross(up, f));
    d3 u = cross(f, s);
    m4 r = identity4();
    r.c[0].x = s.x; r.c[1].x = s.y; r.c[2].x = s.z;
    r.c[0].y = u.x; r.c[1].y = u.y; r.c[2].y = u.z;
    r.c[0].z = f.x; r.c[1].z = f.y; r.c[2].z = f.z;
    r.c[3].x = -dot(s, eye); r.c[3].y = -dot(u, eye); r.c[3].z = -dot(f, eye);
    return r;
}

}  // namespace ub
