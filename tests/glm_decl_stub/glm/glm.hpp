// Declarations only, for a -fsyntax-only check of the URenderer adapter against the reference's headers
// (tests/test_adapter_compiles.py). GLM is a third-party dependency of the reference that this repository does
// not ship; these are NOT an implementation of it: no operator, no function, nothing here computes anything.
// Only the type names the reference's interface headers mention exist, with the member layout the adapter's
// static checks rely on (glm::dvec3 = three contiguous doubles).
#pragma once
namespace glm {
struct dvec2 { double x, y; };
struct dvec3 { double x, y, z; };
struct dvec4 { double x, y, z, w; };
struct dmat3x3 { dvec3 c[3]; };
struct dmat4x4 { dvec4 c[4]; };
struct vec3 { float x, y, z; };
struct mat3x3 { vec3 c[3]; };
}  // namespace glm
