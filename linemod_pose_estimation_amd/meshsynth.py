"""Rendered views of the reference's own meshes: realistic template banks and scenes (test / bench plumbing, not product).

The reference trains its banks by rendering an STL mesh from a grid of view points and feeding image + mask (+ depth) to
`Detector::addTemplate` (/root/reference/src/renderer_only_image.cpp:127-319, src/renderer.cpp:262-308; parameters in
launch/start_object_renderer.launch).  Its real banks are absent (/root/reference/.MISSING_LARGE_BLOBS) but the meshes
(config/stl/memoryChip2.stl, cpu_binary.stl) and one complete pose list (config/data/boxNew_longDistance_linemod_xtion_
renderer_params.yml: 26 view directions x 6 distances x 17 in-plane rotations = 2652 templates) are not.  This module renders
those meshes from that view grid with a small z-buffer rasteriser (meshraster.c, plain C built with gcc) so that a bank consists of NEIGHBOURING
VIEWS OF ONE OBJECT -- unlike synth.make_bank's independent random contours -- and scenes contain rendered instances of the
same object on a textured background.  The OpenGL renderer itself (ork_renderer) stays out of scope.

Fixtures (tests/golden/meshes/, made by tests/golden/make_mesh_fixtures.py from the reference's data files):
  <mesh>.npz     triangles float32 [n, 3, 3], metres, object frame
  views.npz      R float64 [442, 3, 3] (object -> camera rotations of the reference's pose list, one distance ring),
                 direction int32 [442] (index of the view direction), order note
"""
import math
import os

import numpy as np

from .bank import DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL, TemplateBank

MESH_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "meshes")
# launch/start_object_renderer.launch: the ensenso camera of the memory-chip / cpu set-ups
ENSENSO = {"fx": 826.119324, "fy": 826.119324, "radius_min": 0.4, "radius_max": 0.65, "radius_step": 0.05}


def load_mesh(name):
    return np.load(os.path.join(MESH_DIR, name + ".npz"))["triangles"].astype(np.float64)


def load_views():
    d = np.load(os.path.join(MESH_DIR, "views.npz"))
    return d["R"].astype(np.float64), d["direction"].astype(np.int32)


def view_grid(radii=None):
    """The reference's iteration order (pose list of the params yml): view direction -> distance -> in-plane rotation.
    -> list of (R [3,3], distance in metres)."""
    R, direction = load_views()
    if radii is None:
        n = int(round((ENSENSO["radius_max"] - ENSENSO["radius_min"]) / ENSENSO["radius_step"])) + 1
        radii = [ENSENSO["radius_min"] + i * ENSENSO["radius_step"] for i in range(n)]
    out = []
    for d in np.unique(direction):
        idx = np.nonzero(direction == d)[0]
        for r in radii:
            for i in idx:
                out.append((R[i], float(r)))
    return out


_HERE = os.path.dirname(os.path.abspath(__file__))
_RASTER_SRC = os.path.join(_HERE, "meshraster.c")
_RASTER_SO = os.path.join(_HERE, "libmeshraster.so")
_raster = None


def build_raster(force=False):
    """gcc build of the rasteriser (plain C, deterministic double arithmetic: -ffp-contract=off, no -ffast-math)."""
    import subprocess
    if force or not os.path.exists(_RASTER_SO) or os.path.getmtime(_RASTER_SO) < os.path.getmtime(_RASTER_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _RASTER_SO, _RASTER_SRC, "-lm"])
    return _RASTER_SO


def _raster_lib():
    global _raster
    if _raster is None:
        import ctypes as C
        _raster = C.CDLL(build_raster())
        dp = C.POINTER(C.c_double)
        _raster.meshraster_render.argtypes = [dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp]
        _raster.meshraster_render.restype = C.c_int
    return _raster


def render_view(tri, R, distance, fx, fy, width, height, cx=None, cy=None, light=(0.35, -0.45, -0.82)):
    """Z-buffer rendering of `tri` (float64 [n,3,3], object frame, metres) seen with X_cam = R X_obj + (0, 0, distance)
    (meshraster.c).  -> (gray u8 [H,W] (0 outside the object), depth u16 mm [H,W] (0 outside), mask u8 [H,W] (255 on the object),
    rect (x, y, w, h) of the silhouette).  Shading: Lambertian, two-sided, ambient 0.25, gray = 40 + 190 * shade."""
    import ctypes as C
    cx = width / 2.0 if cx is None else cx
    cy = height / 2.0 if cy is None else cy
    tri = np.ascontiguousarray(tri, np.float64)
    Rm = np.ascontiguousarray(R, np.float64)
    lt = np.ascontiguousarray(light, np.float64)
    z = np.empty((height, width), np.float64)
    sh = np.empty((height, width), np.float64)
    dp = C.POINTER(C.c_double)
    n = _raster_lib().meshraster_render(tri.ctypes.data_as(dp), tri.shape[0], Rm.ctypes.data_as(dp), float(distance), float(fx), float(fy), float(cx), float(cy),
                                        width, height, lt.ctypes.data_as(dp), z.ctypes.data_as(dp), sh.ctypes.data_as(dp))
    if n < 0:
        raise ValueError("meshsynth.render_view: the mesh reaches behind the camera at distance %g" % distance)
    cov = np.isfinite(z)
    gray = np.where(cov, np.clip(np.rint(40.0 + 190.0 * sh), 0, 255), 0).astype(np.uint8)
    depth = np.where(cov, np.clip(np.rint(np.where(cov, z, 0.0) * 1000.0), 0, 65535), 0).astype(np.uint16)
    mask = cov.astype(np.uint8) * 255
    ys, xs = np.nonzero(cov)
    rect = (int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)) if len(xs) else (0, 0, 0, 0)
    return gray, depth, mask, rect


def training_view(tri, R, distance, width=640, height=480, fx=ENSENSO["fx"], fy=ENSENSO["fy"], background=0):
    """What the reference's trainer hands addTemplate for one pose (src/renderer.cpp:288-308): -> (bgr u8 [H,W,3], depth u16 mm [H,W],
    mask u8 [H,W], rect).  The colour image is the gray rendering on a black background replicated to three channels."""
    gray, depth, mask, rect = render_view(tri, R, distance, fx, fy, width, height)
    if background:
        gray = np.where(mask > 0, gray, np.uint8(background)).astype(np.uint8)
    bgr = np.ascontiguousarray(np.repeat(gray[:, :, None], 3, 2))
    return bgr, depth, mask, rect


def train_bank(add_template, tri, views, modalities=("ColorGradient", "DepthNormal"), width=640, height=480, class_id="obj", progress=None):
    """Renders every (R, distance) of `views` and calls add_template(sources, class_id, mask) -> (template_id, bbox) -- a
    NativeBank.add_template (HIP trainer) or an OracleDetector.add_template (CPU).  -> list of dicts per ACCEPTED template:
    {view (index into views), rect (x, y, w, h of the silhouette), distance}: the renderer-params side-car (Rects_, Origin_dists_)."""
    meta = []
    for i, (R, dist) in enumerate(views):
        bgr, depth, mask, rect = training_view(tri, R, dist, width, height)
        sources = [bgr if m == "ColorGradient" else depth for m in modalities]
        tid, _ = add_template(sources, class_id, mask)
        if tid >= 0:
            meta.append({"view": i, "rect": rect, "distance": dist})
        if progress and (i + 1) % progress == 0:
            print("meshsynth.train_bank: %d / %d views, %d templates" % (i + 1, len(views), len(meta)), flush=True)
    return meta


def load_bank(name="memoryChip2"):
    """The committed mesh-rendered bank (tests/golden/mesh_bank_<name>.npz, made by tests/golden/make_mesh_bank.py with the oracle's
    trainer) -> (TemplateBank, rects int32 [n,4], distances float64 [n], views int32 [n])."""
    z = np.load(os.path.join(os.path.dirname(MESH_DIR), "mesh_bank_%s.npz" % name))
    bank = empty_bank(tuple(str(m) for m in z["modalities"]), tuple(int(t) for t in z["T"]))
    bank.classes.append(("obj", z["templates"].astype(np.int32), z["features"].astype(np.int32)))
    return bank, z["rects"].astype(np.int32), z["distances"].astype(np.float64), z["views"].astype(np.int32)


def load_banks(names=("memoryChip2", "cpu_binary")):
    """Several committed banks as ONE detector bank, one class per mesh, the class id = the mesh's name (BASELINE configs[2]:
    "2 objects (memoryChip2 + cpu_binary)") -> (TemplateBank, {name: (rects, distances, views)})."""
    bank, side = None, {}
    for name in names:
        b, rects, dists, views = load_bank(name)
        cls = (name,) + tuple(b.classes[0][1:])
        if bank is None:
            bank = b
            bank.classes[0] = cls
        else:
            assert bank.T == b.T and [m["type"] for m in bank.modalities] == [m["type"] for m in b.modalities]
            bank.classes.append(cls)
        side[name] = (rects, dists, views)
    return bank, side


def empty_bank(modalities=("ColorGradient", "DepthNormal"), T=(5, 8)):
    mods = [dict(DEFAULT_COLOR_GRADIENT if m == "ColorGradient" else DEFAULT_DEPTH_NORMAL) for m in modalities]
    return TemplateBank(T=list(T), modalities=mods)


def _smooth_noise(rng, H, W, sigma_px, amp):
    from .synth import _smooth_noise as sn
    return sn(rng, H, W, sigma_px, amp)


def make_scene(tri, views, width=640, height=480, seed=0, n_instances=3, fx=ENSENSO["fx"], fy=ENSENSO["fy"], texture=0.6, depth=True,
               other_tri=None, n_other=0, margin=48, other_class=None):
    """A table-top scene: textured background + tilted plane depth + `n_instances` rendered instances of the mesh, each at a pose of
    the training grid (so exact-pose true positives exist) pasted at a random image position, plus `n_other` instances of another
    mesh: distractors, or -- with `other_class` -- instances of a second trained object that are listed in `truth` under that class
    (the first mesh's are listed under "obj").  -> (sources [bgr u8 HxWx3, depth u16 mm HxW], truth list of {view, x, y, class})."""
    rng = np.random.default_rng([seed, 7000])
    H, W = height, width
    base = rng.uniform(140, 200, 3)
    tex = _smooth_noise(rng, H, W, 8, 8.0 * texture)
    img = np.empty((H, W, 3), np.float64)
    for c in range(3):
        img[:, :, c] = base[c] + tex + _smooth_noise(rng, H, W, 16, 5.0 * texture)
    ys, xs = np.mgrid[0:H, 0:W]
    phi = rng.uniform(0, 2 * np.pi)
    dimg = rng.uniform(700, 850) + 0.25 * (math.cos(phi) * (xs - W / 2) + math.sin(phi) * (ys - H / 2))
    truth, placed = [], []
    jobs = [(tri, "obj")] * n_instances + ([(other_tri, other_class)] * n_other if other_tri is not None else [])
    for mesh, cls in jobs:
        vi = int(rng.integers(0, len(views)))
        R, dist = views[vi]
        gray, d, mask, rect = render_view(mesh, R, dist, fx, fy, W, H)
        x, y, w, h = rect
        # upstream's refinement clamps a match into [8T, size - template - 8T]: an object closer than 8 * T0 = 40 px to the
        # border cannot be reported at its own position, so the generator keeps `margin` px free
        if w == 0 or w + 2 * margin + 1 >= W or h + 2 * margin + 1 >= H:
            continue
        for _ in range(30):   # no overlaps: a later instance must not cover an earlier one
            tx = int(rng.integers(margin, W - w - margin)) - x
            ty = int(rng.integers(margin, H - h - margin)) - y
            box = (x + tx - 6, y + ty - 6, x + tx + w + 6, y + ty + h + 6)
            if all(box[2] <= b[0] or b[2] <= box[0] or box[3] <= b[1] or b[3] <= box[1] for b in placed):
                break
        else:
            continue
        placed.append(box)
        sub = mask[y:y + h, x:x + w] > 0
        gain = rng.uniform(0.75, 1.0)
        for c in range(3):
            dst = img[y + ty:y + ty + h, x + tx:x + tx + w, c]
            dst[sub] = gray[y:y + h, x:x + w][sub] * gain
        zoff = dimg[y + ty:y + ty + h, x + tx:x + tx + w][sub].min() - 8.0 - d[y:y + h, x:x + w][sub].astype(np.float64).max()
        dd = dimg[y + ty:y + ty + h, x + tx:x + tx + w]
        dd[sub] = d[y:y + h, x:x + w][sub].astype(np.float64) + zoff
        if cls:
            truth.append({"view": vi, "x": x + tx, "y": y + ty, "class": cls})
    noise = rng.normal(0, 1.5, (H, W, 3))
    bgr = np.clip(np.rint(img + noise), 0, 255).astype(np.uint8)
    d16 = np.clip(np.rint(dimg + rng.normal(0, 0.3, (H, W))), 1, 65535).astype(np.uint16)
    d16[rng.uniform(0, 1, (H, W)) < 0.01] = 0
    return ([bgr, d16] if depth else [bgr]), truth
