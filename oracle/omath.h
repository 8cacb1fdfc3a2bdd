/*
 * omath.h — TEST INFRASTRUCTURE (CPU oracle), not part of the shipped backend.
 *
 * Vector / matrix arithmetic of the reference restated in plain C++.  The reference takes
 * this arithmetic from GLM (header-only third-party library, included at uengine/umath.h:4-12,
 * no version pinned anywhere in the reference tree, and absent from this build environment).
 * The operation ORDER below follows GLM 0.9.8's generic (non-SIMD) code paths so that a real-GLM
 * build of the reference produces the same bits.  PARITY UNPINNED: the reference ships no test,
 * golden vector or fixture for any of these functions, so nothing checks this restatement
 * against GLM itself.
 *
 * Compile with -ffp-contract=off (the reference is x86-64 g++ without FMA contraction).
 */
#ifndef ORACLE_OMATH_H
#define ORACLE_OMATH_H

#include <cmath>

namespace om {

struct dvec2 { double x, y; };
struct dvec3 { double x, y, z; };
struct dvec4 { double x, y, z, w; };
struct fvec3 { float x, y, z; };
/* column-major: c[i] is column i, as glm::dmat4x4 m[i] */
struct dmat4 { dvec4 c[4]; };
struct dmat3 { dvec3 c[3]; };
struct fmat3 { fvec3 c[3]; };

inline dvec3 operator+(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(dvec3 a, dvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator-(dvec3 a) { return {-a.x, -a.y, -a.z}; }
inline dvec3 operator*(dvec3 a, dvec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline dvec3 operator*(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(double s, dvec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline dvec3 operator/(dvec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline dvec3& operator+=(dvec3& a, dvec3 b) { a = a + b; return a; }
inline dvec3& operator*=(dvec3& a, double s) { a = a * s; return a; }
inline dvec3& operator/=(dvec3& a, double s) { a = a / s; return a; }

inline dvec4 operator+(dvec4 a, dvec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline dvec4 operator*(dvec4 a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline dvec4 operator*(dvec4 a, dvec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline dvec4 operator-(dvec4 a, dvec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

inline dvec3 v3(dvec4 v) { return {v.x, v.y, v.z}; }
inline dvec4 v4(dvec3 v, double w) { return {v.x, v.y, v.z, w}; }

/* glm::dot: component products summed left to right */
inline double dot(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* glm::dot(vec4): (x+y)+(z+w) */
inline double dot(dvec4 a, dvec4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
inline dvec3 cross(dvec3 x, dvec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline double length(dvec3 a) { return std::sqrt(dot(a, a)); }
inline double distance(dvec3 p0, dvec3 p1) { return length(p1 - p0); }
/* glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
inline dvec3 normalize(dvec3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }
/* glm::reflect: I - N * dot(N, I) * 2 */
inline dvec3 reflect(dvec3 I, dvec3 N) { return I - N * dot(N, I) * 2.0; }

/* mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*w) */
inline dvec4 operator*(const dmat4& m, dvec4 v) {
    dvec4 a0 = m.c[0] * v.x + m.c[1] * v.y;
    dvec4 a1 = m.c[2] * v.z + m.c[3] * v.w;
    return a0 + a1;
}
/* vec4 * mat4 (row vector): component i = m[i].x*v.x + m[i].y*v.y + m[i].z*v.z + m[i].w*v.w */
inline dvec4 operator*(dvec4 v, const dmat4& m) {
    return {m.c[0].x * v.x + m.c[0].y * v.y + m.c[0].z * v.z + m.c[0].w * v.w,
            m.c[1].x * v.x + m.c[1].y * v.y + m.c[1].z * v.z + m.c[1].w * v.w,
            m.c[2].x * v.x + m.c[2].y * v.y + m.c[2].z * v.z + m.c[2].w * v.w,
            m.c[3].x * v.x + m.c[3].y * v.y + m.c[3].z * v.z + m.c[3].w * v.w};
}
/* mat3 * vec3 */
inline dvec3 operator*(const dmat3& m, dvec3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
/* vec3 * mat3 (row vector) = transpose(m) * v */
inline dvec3 operator*(dvec3 v, const dmat3& m) {
    return {m.c[0].x * v.x + m.c[0].y * v.y + m.c[0].z * v.z,
            m.c[1].x * v.x + m.c[1].y * v.y + m.c[1].z * v.z,
            m.c[2].x * v.x + m.c[2].y * v.y + m.c[2].z * v.z};
}
/* float mat3 * float vec3 (the glm::mat3x3 TNB of ubdptrenderer.cpp:287,448) */
inline fvec3 operator*(const fmat3& m, fvec3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
inline fvec3 to_f(dvec3 v) { return {(float)v.x, (float)v.y, (float)v.z}; }
inline dvec3 to_d(fvec3 v) { return {(double)v.x, (double)v.y, (double)v.z}; }

inline dmat4 identity4() {
    return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
inline dmat4 operator*(const dmat4& m, double s) {
    return {{m.c[0] * s, m.c[1] * s, m.c[2] * s, m.c[3] * s}};
}
/* glm::translate: m[3] = m0*x + m1*y + m2*z + m3 */
inline dmat4 translate(const dmat4& m, dvec3 v) {
    dmat4 r = m;
    r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
    return r;
}
inline dmat4 scale(const dmat4& m, dvec3 v) {
    return {{m.c[0] * v.x, m.c[1] * v.y, m.c[2] * v.z, m.c[3]}};
}

/* glm::inverse(mat4): cofactor expansion, multiplied by the reciprocal determinant */
inline dmat4 inverse(const dmat4& mm) {
    const dvec4* m = mm.c;
#define E(i, j) (((const double*)&m[i])[j])
    double Coef00 = E(2, 2) * E(3, 3) - E(3, 2) * E(2, 3);
    double Coef02 = E(1, 2) * E(3, 3) - E(3, 2) * E(1, 3);
    double Coef03 = E(1, 2) * E(2, 3) - E(2, 2) * E(1, 3);
    double Coef04 = E(2, 1) * E(3, 3) - E(3, 1) * E(2, 3);
    double Coef06 = E(1, 1) * E(3, 3) - E(3, 1) * E(1, 3);
    double Coef07 = E(1, 1) * E(2, 3) - E(2, 1) * E(1, 3);
    double Coef08 = E(2, 1) * E(3, 2) - E(3, 1) * E(2, 2);
    double Coef10 = E(1, 1) * E(3, 2) - E(3, 1) * E(1, 2);
    double Coef11 = E(1, 1) * E(2, 2) - E(2, 1) * E(1, 2);
    double Coef12 = E(2, 0) * E(3, 3) - E(3, 0) * E(2, 3);
    double Coef14 = E(1, 0) * E(3, 3) - E(3, 0) * E(1, 3);
    double Coef15 = E(1, 0) * E(2, 3) - E(2, 0) * E(1, 3);
    double Coef16 = E(2, 0) * E(3, 2) - E(3, 0) * E(2, 2);
    double Coef18 = E(1, 0) * E(3, 2) - E(3, 0) * E(1, 2);
    double Coef19 = E(1, 0) * E(2, 2) - E(2, 0) * E(1, 2);
    double Coef20 = E(2, 0) * E(3, 1) - E(3, 0) * E(2, 1);
    double Coef22 = E(1, 0) * E(3, 1) - E(3, 0) * E(1, 1);
    double Coef23 = E(1, 0) * E(2, 1) - E(2, 0) * E(1, 1);

    dvec4 Fac0{Coef00, Coef00, Coef02, Coef03};
    dvec4 Fac1{Coef04, Coef04, Coef06, Coef07};
    dvec4 Fac2{Coef08, Coef08, Coef10, Coef11};
    dvec4 Fac3{Coef12, Coef12, Coef14, Coef15};
    dvec4 Fac4{Coef16, Coef16, Coef18, Coef19};
    dvec4 Fac5{Coef20, Coef20, Coef22, Coef23};

    dvec4 Vec0{E(1, 0), E(0, 0), E(0, 0), E(0, 0)};
    dvec4 Vec1{E(1, 1), E(0, 1), E(0, 1), E(0, 1)};
    dvec4 Vec2{E(1, 2), E(0, 2), E(0, 2), E(0, 2)};
    dvec4 Vec3{E(1, 3), E(0, 3), E(0, 3), E(0, 3)};

    dvec4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
    dvec4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
    dvec4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
    dvec4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;

    dvec4 SignA{+1, -1, +1, -1};
    dvec4 SignB{-1, +1, -1, +1};
    dmat4 Inverse{{Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB}};

    dvec4 Row0{Inverse.c[0].x, Inverse.c[1].x, Inverse.c[2].x, Inverse.c[3].x};
    dvec4 Dot0 = m[0] * Row0;
    double Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    double OneOverDeterminant = 1.0 / Dot1;
#undef E
    return Inverse * OneOverDeterminant;
}

/* glm::lookAtLH */
inline dmat4 lookAtLH(dvec3 eye, dvec3 center, dvec3 up) {
    dvec3 f = normalize(center - eye);
    dvec3 s = normalize(cross(up, f));
    dvec3 u = cross(f, s);
    dmat4 r = identity4();
    r.c[0].x = s.x; r.c[1].x = s.y; r.c[2].x = s.z;
    r.c[0].y = u.x; r.c[1].y = u.y; r.c[2].y = u.z;
    r.c[0].z = f.x; r.c[1].z = f.y; r.c[2].z = f.z;
    r.c[3].x = -dot(s, eye);
    r.c[3].y = -dot(u, eye);
    r.c[3].z = -dot(f, eye);
    return r;
}

} // namespace om
#endif
