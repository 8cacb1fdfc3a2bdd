"""Known-answer tests of the CPU oracle (oracle/bdpt_oracle.cpp) against closed forms and against
plain numpy restatements written here from the reference's formulas. The reference holds no test
vectors of its own, so these closed-form cases are the pins the oracle has ("parity unpinned"
otherwise). Runs without a GPU."""
import math

import numpy as np
import pytest

from util import capi, scenes, ulp_diff

import oracle_py

M64 = (1 << 64) - 1


# ---- counter-based generator (specification shared by oracle and kernels) ---------------------------
def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def uniform_py(seed, pas, pixel, stream, dim):
    k = mix64((seed + 0x9E3779B97F4A7C15 * (pas + 1)) & M64)
    key = mix64(k ^ ((pixel * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & M64))
    ctr = ((stream << 32) | dim) + 1
    h = mix64((key + 0x9E3779B97F4A7C15 * ctr) & M64)
    return (h >> 11) * 2.0 ** -53


def test_uniform_matches_python_restatement():
    for seed, pas, pixel, stream, dim in [(0, 0, 0, 0, 0), (0x5EED, 3, 12345, 1, 17), (M64, 2 ** 40, 2 ** 33 + 5, 1, 8 * 31 + 3),
                                          (1, 255, 512 * 512 - 1, 0, 3)]:
        assert oracle_py.uniform(seed, pas, pixel, stream, dim) == uniform_py(seed, pas, pixel, stream, dim)


def test_uniform_is_in_unit_interval_and_well_spread():
    u = np.array([oracle_py.uniform(7, p, x, s, d) for p in range(8) for x in range(64) for s in (0, 1) for d in range(16)])
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01
    assert abs(np.mean(u < 0.25) - 0.25) < 0.015
    assert len(np.unique(u)) == len(u)


# ---- sampling maps (uengine/umath.cpp:22-86) --------------------------------------------------------
def test_sampling_maps_closed_form():
    rng = np.random.default_rng(0)
    for _ in range(64):
        u1, u2 = rng.random(2)
        # stratified rect: 4 strata -> 2x2 grid, stratum 3 = (1,1)
        r = oracle_py.sample_map(0, 4, 3, u1, u2)
        assert np.allclose(r[:2], [0.5 * (1 + u1), 0.5 * (1 + u2)], rtol=0, atol=1e-16)
        r = oracle_py.sample_map(1, 1, 0, u1, u2)
        assert np.allclose(r[:2], [math.sqrt(u2) * math.cos(2 * math.pi * u1), math.sqrt(u2) * math.sin(2 * math.pi * u1)],
                           rtol=1e-15, atol=1e-16)
        r = oracle_py.sample_map(2, 1, 0, u1, u2)   # uniform hemisphere, y up
        s = math.sqrt(1 - u2 * u2)
        assert np.allclose(r, [math.cos(2 * math.pi * u1) * s, u2, math.sin(2 * math.pi * u1) * s], rtol=1e-15, atol=1e-16)
        assert abs(np.linalg.norm(r) - 1) < 1e-15
        r = oracle_py.sample_map(3, 1, 0, u1, u2)   # cosine hemisphere
        assert np.allclose(r, [math.sqrt(1 - u2) * math.cos(2 * math.pi * u1), math.sqrt(u2),
                               math.sqrt(1 - u2) * math.sin(2 * math.pi * u1)], rtol=1e-15, atol=1e-16)
        r = oracle_py.sample_map(4, 1, 0, u1, u2)   # sphere
        assert abs(np.linalg.norm(r) - 1) < 1e-15 and abs(r[1] - (2 * u2 - 1)) < 1e-16
        r = oracle_py.sample_map(5, 1, 0, u1, u2)   # triangle barycentrics (1-sqrt r1, sqrt r1 (1-r2), sqrt r1 r2)
        s = math.sqrt(u1)
        assert np.allclose(r, [1 - s, s * (1 - u2), s * u2], rtol=1e-15, atol=1e-16) and abs(r.sum() - 1) < 1e-15


# ---- primitives through the scene scan (uengine/ugeometry.cpp:70-114, uengine/uscene.cpp:41-66) ------
def one_sphere(center=(0, 0, 0), radius=1.0):
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -5), (0, 0, 0), 0.7, 1.0))
    b.sphere(center, radius, capi.MAT_LATEX_PAINT, b.color((0.5, 0.5, 0.5)))
    return b.build()


def rays(origins, dirs):
    r = np.zeros(len(origins), capi.RAY_DTYPE)
    r["origin"] = origins
    r["dir"] = dirs
    return r


def test_unit_sphere_known_answers():
    sc = one_sphere()
    orc = oracle_py.Oracle(sc, scenes.make_params(4, 4))
    h = orc.intersect_batch(rays(
        [(0, 0, -5), (0, 0, 0), (0, 0, 5), (0, 1, -5), (0, 1.0000001, -5), (0, 0, -1), (0, 0, 1)],
        [(0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1)]))
    # outside: near root; inside: far root; behind: none; tangent (discriminant 0): far-root branch;
    # just outside: miss; on the surface looking in: the near root is 0 (not > 0), the far root 2 is taken;
    # on the surface looking out: near root -2, far root 0 is not < 0 and is returned (ugeometry.cpp:88-93)
    assert list(h["hit"]) == [1, 1, 0, 1, 0, 1, 1]
    assert h["d"][0] == 4.0 and h["d"][1] == 1.0 and h["d"][3] == 5.0 and h["d"][5] == 2.0 and h["d"][6] == 0.0
    assert np.array_equal(h["Ns"][0], [0, 0, -1]) and np.array_equal(h["Ng"][1], [0, 0, 1])
    assert np.all(h["object"][h["hit"] == 1] == 0) and np.all(h["face"] == -1)   # a sphere has no faces


def test_scaled_sphere_returns_world_ray_parameter():
    # URay::transform does not renormalise the direction (ugeometry.cpp:53-57), so the local root is the world distance
    sc = one_sphere(center=(1, 2, 3), radius=0.25)
    orc = oracle_py.Oracle(sc, scenes.make_params(4, 4))
    h = orc.intersect_batch(rays([(1, 2, -2)], [(0, 0, 1)]))
    assert h["hit"][0] == 1 and abs(h["d"][0] - 4.75) < 1e-14
    # the surface point is reported in object-local space (uapp/object.cpp:14-24): the unit sphere's south pole
    assert np.allclose(h["pos"][0], [0, 0, -1], atol=1e-14)


def tri_scene(tris, W=None):
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -5), (0, 0, 0), 0.7, 1.0))
    tex = b.color((0.5, 0.5, 0.5))
    for t in tris:
        v = np.zeros(3, capi.MESH_VERTEX_DTYPE)
        v["pos"] = t
        v["normal"] = (0, 0, -1)
        v["tangent"] = (1, 0, 0)
        v["uv"] = [(0, 0), (1, 0), (0, 1)]
        b.mesh(v, W if W is not None else scenes.identity(), capi.MAT_LATEX_PAINT, tex)
    return b.build()


def test_triangle_known_answers():
    T = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    orc = oracle_py.Oracle(tri_scene([T]), scenes.make_params(4, 4))
    o = [(0.25, 0.25, -1), (0.5, 0.5, -1), (0, 0, -1), (0.25, 0.25, 1), (0.75, 0.75, -1), (0.25, 0.25, 1), (-1, 0.2, 0)]
    d = [(0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, -1), (1, 0, 0)]
    h = orc.intersect_batch(rays(o, d))
    # interior; on the hypotenuse (u+v == 1 accepted); on a corner (u = v = 0 accepted); behind the origin;
    # outside; from the back side (no culling); in the triangle's plane (a == 0: no valid d,u,v)
    assert list(h["hit"]) == [1, 1, 1, 0, 0, 1, 0]
    assert h["d"][0] == 1.0 and h["tex_u"][0] == 0.25 and h["tex_v"][0] == 0.25
    assert h["d"][1] == 1.0 and h["d"][2] == 1.0 and h["d"][5] == 1.0
    # geometric normal normalize(e1 x e2) = +z, flipped to agree with the shading normal -z (uapp/mesh.cpp:146-150)
    assert np.array_equal(h["Ng"][0], [0, 0, -1])


def test_ties_go_to_first_object_and_lowest_face():
    T = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    orc = oracle_py.Oracle(tri_scene([T, T]), scenes.make_params(4, 4))   # two coincident objects
    h = orc.intersect_batch(rays([(0.25, 0.25, -1)], [(0, 0, 1)]))
    assert h["hit"][0] == 1 and h["object"][0] == 0          # strict '<' in uscene.cpp:52
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -5), (0, 0, 0), 0.7, 1.0))
    v = np.zeros(6, capi.MESH_VERTEX_DTYPE)
    v["pos"] = T + T
    v["normal"] = (0, 0, -1); v["tangent"] = (1, 0, 0)
    b.mesh(v, scenes.identity(), capi.MAT_LATEX_PAINT, b.color((1, 1, 1)))
    orc2 = oracle_py.Oracle(b.build(), scenes.make_params(4, 4))      # one mesh, two coincident faces
    h = orc2.intersect_batch(rays([(0.25, 0.25, -1)], [(0, 0, 1)]))
    assert h["hit"][0] == 1 and h["face"][0] == 0            # strict '<' in uapp/mesh.cpp:132


def test_visibility_has_no_epsilon():
    sc = one_sphere()
    orc = oracle_py.Oracle(sc, scenes.make_params(4, 4))
    seg = np.zeros(4, capi.SEGMENT_DTYPE)
    seg["p0"] = [(0, 0, -5), (0, 0, -5), (0, 3, -5), (0, 0, -5)]
    seg["p1"] = [(0, 0, 5), (0, 0, -2), (0, 3, 5), (0, 0, -1)]
    # through the sphere; stops short of it; passes beside it; ends exactly on it (d == length is not < length)
    assert list(orc.visibility_batch(seg)) == [0, 1, 1, 1]


# ---- derived scene data --------------------------------------------------------------------------------
def test_camera_snapshot():
    sc, params, _ = scenes.make_config("C1")
    orc = oracle_py.Oracle(sc, params)
    V, iV, dist = orc.camera_info()
    V = V.reshape(4, 4).T
    iV = iV.reshape(4, 4).T
    import ctypes
    libm = ctypes.CDLL("libm.so.6")   # the C library's tanf; numpy's float32 tan differs from it in the last bit
    libm.tanf.restype = ctypes.c_float
    libm.tanf.argtypes = [ctypes.c_float]
    assert dist == 1.0 / libm.tanf(0.69 / 2)  # 1.0 / tanf(...), ucamera.cpp:69
    assert np.allclose(V @ iV, np.eye(4), atol=1e-15)
    assert np.allclose(V @ np.array([0, 0, -3.4, 1.0]), [0, 0, 0, 1], atol=1e-15)   # camera position -> view origin
    assert np.allclose(V @ np.array([0, 0, 0, 1.0]), [0, 0, 3.4, 1], atol=1e-15)    # looks down +z (left-handed)


def test_emitter_area_and_probability():
    sc, params, _ = scenes.make_config("C1")
    orc = oracle_py.Oracle(sc, params)
    areas = []
    for i in range(sc.n_objects):
        invW, area, prob, bs = orc.object_info(i)
        if sc.objects[i].is_emitter:
            areas.append((i, area, prob))
    assert len(areas) == 1
    assert abs(areas[0][1] - 0.25) < 1e-6 and areas[0][2] == 1.0   # 0.5 x 0.5 quad, the only emitter


def test_sphere_area_uses_x_axis_scale():
    # ImplicitSphere::area: 4 pi r^2 with r = |W x| (uapp/implicitsphere.cpp:31-38)
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -5), (0, 0, 0), 0.7, 1.0))
    b.sphere((0, 0, 0), 0.5, capi.MAT_LATEX_PAINT, b.color((1, 1, 1)), power=(1, 1, 1))
    orc = oracle_py.Oracle(b.build(), scenes.make_params(4, 4))
    _, area, prob, _ = orc.object_info(0)
    assert abs(area - 4 * math.pi * 0.25) < 1e-14 and prob == 1.0


# ---- BSDFs (uengine/ubsdf*.cpp) ---------------------------------------------------------------------------
def frame_query(lobe, wi, wo, eta=1.5, tex=(0.8, 0.6, 0.4), u1=0.3, u2=0.6):
    q = np.zeros(1, capi.BSDF_QUERY_DTYPE)
    q["lobe"] = lobe; q["eta"] = eta; q["tex"] = tex
    q["frame"][0] = [0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1]   # Ng, Ns = y; Ts = x; Bs = z
    q["wi"] = wi; q["wo"] = wo; q["u1"] = u1; q["u2"] = u2
    return q


def unit(v):
    v = np.asarray(v, float)
    return v / np.linalg.norm(v)


def test_lambertian_value_pdf_and_side_tests():
    up1, up2, down = unit((0.3, 0.8, 0.1)), unit((-0.5, 0.4, 0.2)), unit((0.1, -0.9, 0.2))
    r = oracle_py.bsdf_eval(frame_query(capi.LOBE_LAMBERTIAN, up1, up2))
    assert np.allclose(r["f"][0], np.array([0.8, 0.6, 0.4]) / math.pi, rtol=1e-15)
    assert abs(oracle_py.bsdf_pdf(frame_query(capi.LOBE_LAMBERTIAN, up1, up2))["pdf"][0] - 1 / math.pi) < 1e-16
    # opposite sides: no reflection through the surface
    assert np.all(oracle_py.bsdf_eval(frame_query(capi.LOBE_LAMBERTIAN, up1, down))["f"][0] == 0)
    assert oracle_py.bsdf_pdf(frame_query(capi.LOBE_LAMBERTIAN, up1, down))["pdf"][0] == 0
    s = oracle_py.bsdf_scatter(frame_query(capi.LOBE_LAMBERTIAN, up1, up1, u1=0.25, u2=0.49))
    assert s["ok"][0] == 1 and s["specular"][0] == 0 and abs(s["dir"][0][1] - 0.7) < 1e-15   # y = sqrt(u2)
    assert abs(s["pdf"][0] - 1 / math.pi) < 1e-16


def test_mirror_reflects_and_is_specular():
    w = unit((0.3, 0.8, 0.1))
    s = oracle_py.bsdf_scatter(frame_query(capi.LOBE_MIRROR, w, w))
    assert s["ok"][0] == 1 and s["specular"][0] == 1 and s["pdf"][0] == 1.0
    assert np.allclose(s["dir"][0], [-w[0], w[1], -w[2]], atol=1e-16)
    assert np.allclose(s["f"][0], [0.8, 0.6, 0.4])
    # samplePSA / pPSA of the mirror are NOT delta-aware: same-side pairs return the texture / 1
    # (ubsdfperfectmirror.cpp:18-23, 36-41); the renderer skips specular vertices by their flag instead
    assert np.allclose(oracle_py.bsdf_eval(frame_query(capi.LOBE_MIRROR, w, unit((0, 1, 0))))["f"][0], [0.8, 0.6, 0.4])
    assert oracle_py.bsdf_pdf(frame_query(capi.LOBE_MIRROR, w, unit((0, 1, 0))))["pdf"][0] == 1.0
    assert np.all(oracle_py.bsdf_eval(frame_query(capi.LOBE_MIRROR, w, unit((0, -1, 0))))["f"][0] == 0)


def test_dielectric_fresnel_normal_incidence_and_tir():
    n = (0.0, 1.0, 0.0)
    # u1 < R -> reflect with f = R tex, p = R; normal incidence R = ((1.5-1)/(1.5+1))^2 = 0.04
    s = oracle_py.bsdf_scatter(frame_query(capi.LOBE_DIELECTRIC, n, n, eta=1.5, u1=0.01))
    assert s["ok"][0] == 1 and s["specular"][0] == 1
    assert abs(s["pdf"][0] - 0.04) < 1e-15 and np.allclose(s["dir"][0], [0, 1, 0], atol=1e-15)
    assert np.allclose(s["f"][0], 0.04 * np.array([0.8, 0.6, 0.4]), rtol=1e-13)
    s = oracle_py.bsdf_scatter(frame_query(capi.LOBE_DIELECTRIC, n, n, eta=1.5, u1=0.5))
    assert abs(s["pdf"][0] - 0.96) < 1e-15 and np.allclose(s["dir"][0], [0, -1, 0], atol=1e-15)
    # from inside (w.y < 0) at 60 degrees: sin > 1/1.5 -> total internal reflection, probability 1
    w = np.array([math.sin(math.radians(60)), -math.cos(math.radians(60)), 0.0])
    s = oracle_py.bsdf_scatter(frame_query(capi.LOBE_DIELECTRIC, w, w, eta=1.5, u1=0.999))
    assert s["ok"][0] == 1 and abs(s["pdf"][0] - 1.0) < 1e-15 and s["dir"][0][1] < 0


# ---- image texture (uapp/textureimg.cpp:56-82), numpy restatement ----------------------------------------
def texture_ref(img, u, v):
    """TextureImg::sample as written: u wraps by its own integer part, v wraps by the integer part OF U,
    and the bilinear weights are attached to the opposite texels."""
    h, w = img.shape[:2]
    texel = img[..., :3].astype(np.float64) / 255.0
    u = u - float(int(u))
    v = v - float(int(u))     # after the line above int(u) == 0: v is never wrapped
    x = u * (w - 1)
    y = v * (h - 1)
    x1, x2 = int(math.floor(x)), int(math.ceil(x))
    y1, y2 = int(math.floor(y)), int(math.ceil(y))
    tx, ty = x - x1, y - y1
    c1 = tx * texel[y1, x1] + (1 - tx) * texel[y1, x2]   # tx weighs the floor texel (:78-79)
    c2 = tx * texel[y2, x1] + (1 - tx) * texel[y2, x2]
    return ty * c1 + (1 - ty) * c2


def test_image_texture_against_numpy_restatement():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (7, 5, 4), dtype=np.uint8)
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -5), (0, 0, 0), 0.7, 1.0))
    t = b.image(img)
    b.sphere((0, 0, 0), 1, capi.MAT_LATEX_PAINT, t)
    orc = oracle_py.Oracle(b.build(), scenes.make_params(4, 4))
    uv = np.concatenate([rng.random((64, 2)), [[0, 0], [0.999999, 0.999999], [2.25, 0.5], [0.5, 0.0]]])
    got = orc.texture_sample(t, uv)
    for k in range(len(uv)):
        exp = texture_ref(img, uv[k, 0], uv[k, 1])
        # the oracle follows the source's statement order; the restatement above is the same arithmetic
        assert np.allclose(got[k], exp, rtol=1e-14, atol=1e-16), (k, uv[k], got[k], exp)
