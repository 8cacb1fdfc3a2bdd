"""Procedural, seeded scenes in the flat scene format of include/ubdpt.h.

The reference ships no scene files, so every workload is generated here. The transform helpers
compose matrices the way uapp/scene.cpp does (sphere: scale(translate(I, c), r), :137;
mesh: scale(rotate(translate(I, t), angle, axis), s), :179). Meshes are triangle soups whose
vertex data is float32, like the aiMesh data the reference reads (uapp/mesh.cpp:14-32).
"""
import numpy as np

from . import capi
from .capi import (Camera, FlatScene, MESH_VERTEX_DTYPE, Object, RenderParams, Texture)


# ---- column-major 4x4 helpers (scene-construction side) -------------------------------------------
def identity():
    return np.eye(4)


def translate(m, t):
    r = m.copy()
    r[:, 3] = m[:, 0] * t[0] + m[:, 1] * t[1] + m[:, 2] * t[2] + m[:, 3]
    return r


def scale(m, s):
    s = np.broadcast_to(np.asarray(s, float), (3,))
    r = m.copy()
    r[:, 0] *= s[0]; r[:, 1] *= s[1]; r[:, 2] *= s[2]
    return r


def rotate(m, angle, axis):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    c, s = np.cos(angle), np.sin(angle)
    t = (1 - c) * a
    R = np.eye(4)
    R[0, 0] = c + t[0] * a[0]; R[1, 0] = t[0] * a[1] + s * a[2]; R[2, 0] = t[0] * a[2] - s * a[1]
    R[0, 1] = t[1] * a[0] - s * a[2]; R[1, 1] = c + t[1] * a[1]; R[2, 1] = t[1] * a[2] + s * a[0]
    R[0, 2] = t[2] * a[0] + s * a[1]; R[1, 2] = t[2] * a[1] - s * a[0]; R[2, 2] = c + t[2] * a[2]
    return m @ R


def colmajor(m):
    """4x4 numpy (row, col) -> 16 doubles column-major, as glm stores them."""
    return np.asarray(m, float).T.reshape(16).copy()


def sphere_transform(center, radius):
    return scale(translate(identity(), center), radius)


def mesh_transform(trans=(0, 0, 0), angle=0.0, axis=(1, 1, 1), scl=(1, 1, 1)):
    return scale(rotate(translate(identity(), trans), angle, axis), scl)


# ---- scene assembly --------------------------------------------------------------------------------
class SceneBuilder:
    def __init__(self, camera):
        self.camera = camera
        self.objects = []
        self.textures = []
        self.vertices = []
        self.texels = []
        self.n_faces = 0
        self.texel_bytes = 0

    def color(self, rgb):
        t = Texture()
        t.kind = capi.TEX_COLOR
        t.color[:] = list(map(float, rgb))
        self.textures.append(t)
        return len(self.textures) - 1

    def image(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        assert rgba.ndim == 3 and rgba.shape[2] == 4
        t = Texture()
        t.kind = capi.TEX_IMAGE
        t.height, t.width = rgba.shape[0], rgba.shape[1]
        t.texel_offset = self.texel_bytes
        self.texels.append(rgba.reshape(-1))
        self.texel_bytes += rgba.size
        self.textures.append(t)
        return len(self.textures) - 1

    def _obj(self, kind, W, material, tex, mat_a=0.0, mat_b=0.0, power=None):
        o = Object()
        o.kind = kind
        o.material = material
        o.W[:] = list(colmajor(W))
        o.mat_a, o.mat_b = float(mat_a), float(mat_b)
        o.texture = tex
        o.is_emitter = 1 if power is not None else 0
        if power is not None:
            o.power[:] = list(map(float, power))
        self.objects.append(o)
        return o

    def sphere(self, center, radius, material, tex, W=None, **kw):
        return self._obj(capi.MODEL_SPHERE, W if W is not None else sphere_transform(center, radius), material, tex, **kw)

    def mesh(self, verts, W, material, tex, **kw):
        verts = np.ascontiguousarray(verts, MESH_VERTEX_DTYPE).reshape(-1)
        assert verts.size % 3 == 0 and verts.size > 0
        o = self._obj(capi.MODEL_MESH, W, material, tex, **kw)
        o.first_face = self.n_faces
        o.num_faces = verts.size // 3
        self.n_faces += verts.size // 3
        self.vertices.append(verts)
        return o

    def build(self):
        v = np.concatenate(self.vertices) if self.vertices else np.zeros(0, MESH_VERTEX_DTYPE)
        tx = np.concatenate(self.texels) if self.texels else np.zeros(0, np.uint8)
        return FlatScene(self.camera, self.objects, v, self.textures, tx)


def make_camera(pos, look_at, vfov, ratio):
    c = Camera()
    c.pos[:] = list(map(float, pos))
    c.look_at[:] = list(map(float, look_at))
    c.vfov = float(vfov)
    c.ratio = float(ratio)
    return c


def make_params(w, h, pixel_subdiv=1, lens_subdiv=1, min_depth=5, focus=1.0, lens=0.0001):
    """Defaults are the GUI defaults (uapp/RenderingPage.qml:61-115)."""
    p = RenderParams()
    p.img_res_x, p.img_res_y = w, h
    p.pixel_subdiv, p.lens_subdiv, p.min_depth = pixel_subdiv, lens_subdiv, min_depth
    p.focus_plane_distance, p.lens_size = focus, lens
    return p


# ---- mesh generators -------------------------------------------------------------------------------
def soup_from_indexed(pos, nrm, tan, uv, faces):
    """Indexed mesh -> triangle soup of MeshVertex (3 per face)."""
    faces = np.asarray(faces, np.int64)
    out = np.zeros(faces.shape[0] * 3, MESH_VERTEX_DTYPE)
    idx = faces.reshape(-1)
    out["pos"] = np.asarray(pos, np.float32)[idx]
    out["normal"] = np.asarray(nrm, np.float32)[idx]
    out["tangent"] = np.asarray(tan, np.float32)[idx]
    out["uv"] = np.asarray(uv, np.float32)[idx]
    return out


def quad(p0, p1, p2, p3, N, T):
    """Two faces (0,1,2),(0,2,3); one normal/tangent for all corners; uv (0,0),(1,0),(1,1),(0,1)."""
    pos = np.array([p0, p1, p2, p3], float)
    nrm = np.tile(np.asarray(N, float), (4, 1))
    tan = np.tile(np.asarray(T, float), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    return soup_from_indexed(pos, nrm, tan, uv, [[0, 1, 2], [0, 2, 3]])


def _grid_faces(nu, nv, wrap_u=True, wrap_v=False):
    """Faces of a (nu x nv) parametric grid of vertices laid out [iv * nu_verts + iu]."""
    nuv = nu if wrap_u else nu + 1
    nvv = nv if wrap_v else nv + 1
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    iu = iu.reshape(-1); iv = iv.reshape(-1)
    iu1 = (iu + 1) % nuv if wrap_u else iu + 1
    iv1 = (iv + 1) % nvv if wrap_v else iv + 1
    a = iv * nuv + iu; b = iv * nuv + iu1; c = iv1 * nuv + iu1; d = iv1 * nuv + iu
    return np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)


def _value_noise(p, seed, freq):
    """Cheap smooth pseudo-noise in [-1,1] from sums of sines with seeded phases/directions."""
    rng = np.random.default_rng(seed)
    out = np.zeros(p.shape[0])
    for k in range(4):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        out += np.sin(p @ d * freq * (1.0 + 0.7 * k) + rng.uniform(0, 2 * np.pi)) / (1.5 ** k)
    return out / 2.5


def displaced_sphere(n_faces_target, seed=1, amp=0.08, freq=5.0):
    """UV sphere with seeded radial displacement; smooth normals by finite differences. Unit scale."""
    nv = max(4, int(np.sqrt(n_faces_target / 4.0)))
    nu = 2 * nv
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    uu = uu.reshape(-1); vv = vv.reshape(-1)

    def surf(uu, vv):
        th = 2 * np.pi * uu
        ph = np.pi * (0.001 + 0.998 * vv)
        d = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], 1)
        return d * (1.0 + amp * _value_noise(d, seed, freq))[:, None]

    e = 1e-4
    P = surf(uu, vv)
    du = (surf(uu + e, vv) - surf(uu - e, vv)) / (2 * e)
    dv = (surf(uu, vv + e) - surf(uu, vv - e)) / (2 * e)
    N = np.cross(du, dv)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    N *= np.sign(np.sum(N * P, 1, keepdims=True))  # outward
    T = du / np.linalg.norm(du, axis=1, keepdims=True)
    T = T - N * np.sum(N * T, 1, keepdims=True)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    faces = _grid_faces(nu, nv, wrap_u=False, wrap_v=False)
    return soup_from_indexed(P, N, T, np.stack([uu, vv], 1), faces)


def torus(n_faces_target, R=1.0, r=0.35, seed=2, amp=0.03, freq=7.0):
    nv = max(4, int(np.sqrt(n_faces_target / 6.0)))
    nu = 3 * nv
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    uu = uu.reshape(-1); vv = vv.reshape(-1)

    def surf(uu, vv):
        th = 2 * np.pi * uu
        ph = 2 * np.pi * vv
        c = np.stack([np.cos(th), np.zeros_like(th), np.sin(th)], 1)
        n = c * np.cos(ph)[:, None] + np.array([0, 1.0, 0])[None, :] * np.sin(ph)[:, None]
        rr = r * (1.0 + amp * _value_noise(c * R + n * r, seed, freq))
        return c * R + n * rr[:, None], n

    e = 1e-4
    P, n0 = surf(uu, vv)
    du = (surf(uu + e, vv)[0] - surf(uu - e, vv)[0]) / (2 * e)
    dv = (surf(uu, vv + e)[0] - surf(uu, vv - e)[0]) / (2 * e)
    N = np.cross(du, dv)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    N *= np.sign(np.sum(N * n0, 1, keepdims=True))
    T = du / np.linalg.norm(du, axis=1, keepdims=True)
    T = T - N * np.sum(N * T, 1, keepdims=True)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    faces = _grid_faces(nu, nv, wrap_u=False, wrap_v=False)
    return soup_from_indexed(P, N, T, np.stack([uu * 4, vv], 1), faces)


def checker_texture(size=512, seed=3, cells=8):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]
    chk = ((x * cells // size) + (y * cells // size)) % 2
    base = np.where(chk[..., None] == 1, np.array([230, 225, 215]), np.array([60, 90, 150]))
    noise = rng.integers(-20, 21, size=(size, size, 1))
    rgb = np.clip(base + noise, 0, 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((size, size, 1), 255, np.uint8)], 2)


# ---- the configurations ----------------------------------------------------------------------------
GREY, RED, GREEN = (0.725, 0.725, 0.725), (0.63, 0.065, 0.05), (0.14, 0.45, 0.091)


def _cornell_shell(b, left=None, right=None, grey=None):
    """Five quad walls + the quad emitter, identity transforms. Object order: floor, ceiling, back,
    left, right, emitter."""
    I = identity()
    grey = b.color(GREY) if grey is None else grey
    left = b.color(RED) if left is None else left
    right = b.color(GREEN) if right is None else right
    LP = capi.MAT_LATEX_PAINT
    b.mesh(quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), (0, 1, 0), (1, 0, 0)), I, LP, grey)     # floor
    b.mesh(quad((-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1), (0, -1, 0), (1, 0, 0)), I, LP, grey)         # ceiling
    b.mesh(quad((-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1), (0, 0, -1), (1, 0, 0)), I, LP, grey)         # back
    b.mesh(quad((-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (1, 0, 0), (0, 0, 1)), I, LP, left)      # left
    b.mesh(quad((1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1), (-1, 0, 0), (0, 0, 1)), I, LP, right)        # right
    b.mesh(quad((-.25, .999, -.25), (-.25, .999, .25), (.25, .999, .25), (.25, .999, -.25), (0, -1, 0), (1, 0, 0)),
           I, LP, b.color((0.8, 0.8, 0.8)), power=(20, 20, 20))                                               # emitter


def _cornell_camera(ratio=1.0):
    return make_camera((0, 0, -3.4), (0, 0, 0), 0.69, ratio)


def cornell_c1():
    """Cornell box, two LatexPaint implicit spheres, one quad emitter."""
    b = SceneBuilder(_cornell_camera())
    _cornell_shell(b)
    white = b.color((0.8, 0.8, 0.8))
    b.sphere((-0.4, -0.65, 0.3), 0.35, capi.MAT_LATEX_PAINT, white)
    b.sphere((0.45, -0.65, -0.2), 0.35, capi.MAT_LATEX_PAINT, white)
    return b.build()


def cornell_c2():
    """Same box, Dielectric(eta 1.5, tex 1) and PerfectMirror(tex 0.9) spheres (caustics)."""
    b = SceneBuilder(_cornell_camera())
    _cornell_shell(b)
    b.sphere((-0.4, -0.65, 0.3), 0.35, capi.MAT_DIELECTRIC, b.color((1, 1, 1)), mat_a=1.5)
    b.sphere((0.45, -0.65, -0.2), 0.35, capi.MAT_PERFECT_MIRROR, b.color((0.9, 0.9, 0.9)))
    return b.build()


def mesh_scene(n_tris=100_000, seed=1, ratio=16.0 / 9.0, mixed=False):
    """Box + displaced sphere + torus (+ small sphere) tessellated to about n_tris triangles,
    one image texture and colour textures; `mixed` adds Glossy / mirror / dielectric materials."""
    b = SceneBuilder(_cornell_camera(ratio))
    chk = b.image(checker_texture(512, seed + 2))
    _cornell_shell(b, grey=None)
    n_big = int(n_tris * 0.6)
    n_tor = int(n_tris * 0.3)
    n_small = max(64, n_tris - n_big - n_tor - 12)
    W1 = mesh_transform((-0.35, -0.55, 0.25), 0.4, (0, 1, 0), (0.42, 0.42, 0.42))
    W2 = mesh_transform((0.45, -0.7, -0.15), 0.9, (1, 0.3, 0), (0.3, 0.3, 0.3))
    W3 = mesh_transform((0.1, 0.35, 0.5), 0.0, (1, 1, 1), (0.22, 0.3, 0.22))
    LP = capi.MAT_LATEX_PAINT
    b.mesh(displaced_sphere(n_big, seed), W1, capi.MAT_GLOSSY if mixed else LP, chk, mat_a=0.6, mat_b=0.3)
    b.mesh(torus(n_tor, seed=seed + 1), W2, capi.MAT_PERFECT_MIRROR if mixed else LP, b.color((0.75, 0.6, 0.3)))
    b.mesh(displaced_sphere(n_small, seed + 5, amp=0.15, freq=3.0), W3, LP, b.color((0.3, 0.5, 0.8)))
    if mixed:
        b.sphere((-0.55, 0.1, -0.3), 0.2, capi.MAT_DIELECTRIC, b.color((1, 1, 1)), mat_a=1.5)
    return b.build()


def tiny_scene(seed=0):
    """Small mixed scene for edge-case parity: rotated / non-uniformly scaled meshes, a sphere
    emitter, a mesh emitter, every material."""
    b = SceneBuilder(make_camera((0.2, 0.3, -3.0), (0, -0.1, 0), 0.8, 4.0 / 3.0))
    chk = b.image(checker_texture(16, seed + 1, cells=4))
    _cornell_shell(b, grey=chk)
    b.mesh(displaced_sphere(200, seed + 2), mesh_transform((-0.4, -0.5, 0.2), 0.7, (1, 2, 0.5), (0.3, 0.45, 0.3)),
           capi.MAT_GLOSSY, b.color((0.7, 0.7, 0.2)), mat_a=0.5, mat_b=0.4)
    b.mesh(torus(150, seed=seed + 3), mesh_transform((0.4, -0.6, -0.1), 1.1, (0.2, 1, 0.1), (0.3, 0.3, 0.3)),
           capi.MAT_DIELECTRIC, b.color((0.95, 1.0, 0.95)), mat_a=1.33)
    b.sphere((0.0, -0.75, 0.5), 0.25, capi.MAT_PERFECT_MIRROR, b.color((0.9, 0.9, 0.9)))
    b.sphere((0.6, 0.5, 0.4), 0.08, capi.MAT_LATEX_PAINT, b.color((0.8, 0.8, 0.8)), power=(3, 2.5, 2))
    return b.build()


def many_spheres_scene(n_spheres, seed=0):
    """Cornell shell plus a seeded cloud of small spheres of every material: a scene whose OBJECT count
    (6 + n_spheres), not its triangle count, is the variable."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder(_cornell_camera())
    _cornell_shell(b)
    mats = [(capi.MAT_LATEX_PAINT, 0.0, 0.0), (capi.MAT_GLOSSY, 0.5, 0.3), (capi.MAT_PERFECT_MIRROR, 0.0, 0.0),
            (capi.MAT_DIELECTRIC, 1.5, 0.0)]
    for k in range(n_spheres):
        c = rng.uniform(-0.8, 0.8, 3)
        r = float(rng.uniform(0.04, 0.12))
        m, a, bb = mats[k % len(mats)]
        col = (1, 1, 1) if m == capi.MAT_DIELECTRIC else tuple(rng.uniform(0.3, 0.9, 3))
        b.sphere(tuple(c), r, m, b.color(col), mat_a=a, mat_b=bb)
    return b.build()


CONFIGS = {
    # name: (scene factory, width, height, spp)
    "C1": (cornell_c1, 256, 256, 16),
    "C2": (cornell_c2, 512, 512, 256),
    "C3": (lambda: mesh_scene(100_000, 1), 1920, 1080, 64),
    "C4": (lambda: mesh_scene(2_000_000, 1), 1920, 1080, 256),
    "C5": (lambda: mesh_scene(100_000, 1, mixed=True), 3840, 2160, 4096),
}


def make_config(name):
    factory, w, h, spp = CONFIGS[name]
    scene = factory()
    scene.desc.camera.ratio = float(w) / float(h)
    return scene, make_params(w, h), spp
