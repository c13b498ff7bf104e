"""Seeded synthetic template banks and RGB-D scenes (the reference's real banks are missing:
/root/reference/.MISSING_LARGE_BLOBS; plan: SURVEY.md 8d).

Banks imitate what the reference's trainers write (/root/reference/src/renderer.cpp:262-329): a handful of base
shapes seen at several distances (scale rings), in-plane rotations and tilts, 63 ColorGradient features on the
silhouette at level 0 and 31 at level 1, DepthNormal features inside it, cropped like upstream `cropTemplates`
(SURVEY.md A.11).  Bounding sizes follow the 55-194 px range of the surviving
config/data/boxNew_longDistance_linemod_xtion_renderer_params.yml.  Scenes contain rendered instances of bank
templates (true positives) plus distractors on a textured background, and a tilted-plane depth map.
"""
import math

import numpy as np

from .bank import DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL, TemplateBank


def _convex_base_shape(rng):
    """Unit-scale convex polygon (max radius 1), K vertices on a jittered ellipse."""
    K = int(rng.integers(4, 10))
    ang = np.sort(rng.uniform(0, 2 * np.pi, K))
    # avoid degenerate slivers: spread angles
    ang = (ang + np.linspace(0, 2 * np.pi, K, endpoint=False)) / 2.0
    ang = np.sort(ang % (2 * np.pi))
    ax = rng.uniform(0.55, 1.0)
    pts = np.stack([np.cos(ang), ax * np.sin(ang)], 1)
    return pts / np.abs(pts).max()


def _contour_samples(verts, n):
    """n points evenly spaced by arc length on the closed polygon + outward edge-normal angle (deg) at each."""
    nxt = np.roll(verts, -1, 0)
    seg = nxt - verts
    seglen = np.hypot(seg[:, 0], seg[:, 1])
    cum = np.concatenate([[0.0], np.cumsum(seglen)])
    s = (np.arange(n) + 0.5) * (cum[-1] / n)
    idx = np.minimum(np.searchsorted(cum, s, side="right") - 1, len(verts) - 1)
    t = (s - cum[idx]) / np.maximum(seglen[idx], 1e-9)
    pts = verts[idx] + seg[idx] * t[:, None]
    # polygon is counter-clockwise in (x, y-down) image coords or not: compute signed area to orient normals outward
    area = 0.5 * np.sum(verts[:, 0] * nxt[:, 1] - nxt[:, 0] * verts[:, 1])
    nrm = np.stack([seg[idx, 1], -seg[idx, 0]], 1) * (1.0 if area > 0 else -1.0)
    ang = np.degrees(np.arctan2(nrm[:, 1], nrm[:, 0])) % 360.0
    return pts, ang, cum[-1]


def _angle_to_label(ang_deg):
    """Same folding as upstream hysteresisGradient: 16 bins over 360 deg, then & 7 (orientation mod 180)."""
    return (np.rint(ang_deg * (16.0 / 360.0)).astype(np.int64) & 7).astype(np.int32)


def make_bank(n_templates, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=0, class_id="obj",
              num_features=63, size_range=(55.0, 194.0), n_rings=6, classes=None):
    """Build a TemplateBank with `n_templates` pyramids per class.  `classes` = list of class ids (default [class_id]).
    bank.meta[class_id] = list of dicts(verts (float Kx2, level-0 px, relative to crop origin), plane_label, width, height)."""
    T = list(T)
    L = len(T)
    mods = []
    for m in modalities:
        d = dict(DEFAULT_COLOR_GRADIENT if m == "ColorGradient" else DEFAULT_DEPTH_NORMAL)
        d["num_features"] = num_features
        mods.append(d)
    M = len(mods)
    bank = TemplateBank(T=T, modalities=mods)
    for ci, cid in enumerate(classes or [class_id]):
        rng = np.random.default_rng([seed, ci])
        n_base = max(1, n_templates // 384)
        bases = [_convex_base_shape(rng) for _ in range(n_base)]
        templates = np.zeros((n_templates * L * M, 5), np.int32)
        feats = []
        meta = []
        fbegin = 0
        for i in range(n_templates):
            base = bases[i % n_base]
            ring = (i // n_base) % n_rings
            size = size_range[0] + (size_range[1] - size_range[0]) * (ring + rng.uniform(0, 1)) / n_rings
            theta = rng.uniform(0, 2 * np.pi)
            tilt = rng.uniform(0.5, 1.0)
            A = np.array([[math.cos(theta), -math.sin(theta)], [math.sin(theta), math.cos(theta)]]) @ np.diag([1.0, tilt])
            verts = (base @ A.T)
            verts = verts / np.abs(verts).max() * (size / 2.0)
            plane_label = int(rng.integers(0, 8))
            per_level = []  # [l][m] -> (x, y, label) arrays in level coordinates (uncropped, centred)
            for l in range(L):
                nf = max(1, num_features >> l)
                lv = []
                for m in mods:
                    if m["type"] == "ColorGradient":
                        pts, ang, _ = _contour_samples(verts, nf * 4)
                        sel = (np.arange(nf) * 4 + int(rng.integers(0, 4))) % (nf * 4)
                        p = np.floor(pts[sel] / (1 << l)).astype(np.int32)
                        lab = _angle_to_label(ang[sel])
                    else:
                        pts, _, _ = _contour_samples(verts, nf)
                        t = rng.uniform(0.1, 0.8, nf)[:, None]
                        p = np.floor(pts * t / (1 << l)).astype(np.int32)
                        lab = np.full(nf, plane_label, np.int32)
                        noisy = rng.uniform(0, 1, nf) < 0.15
                        lab[noisy] = rng.integers(0, 8, int(noisy.sum()))
                    lv.append((p[:, 0].copy(), p[:, 1].copy(), lab))
                per_level.append(lv)
            # cropTemplates (A.11): bounding box over all levels in level-0 units, even origin
            min_x = min(int((x << l).min()) for l, lv in enumerate(per_level) for (x, y, _) in lv)
            min_y = min(int((y << l).min()) for l, lv in enumerate(per_level) for (x, y, _) in lv)
            max_x = max(int((x << l).max()) for l, lv in enumerate(per_level) for (x, y, _) in lv)
            max_y = max(int((y << l).max()) for l, lv in enumerate(per_level) for (x, y, _) in lv)
            if min_x % 2 == 1:
                min_x -= 1
            if min_y % 2 == 1:
                min_y -= 1
            for l, lv in enumerate(per_level):
                for m_i, (x, y, lab) in enumerate(lv):
                    k = (i * L + l) * M + m_i
                    f = np.stack([x - (min_x >> l), y - (min_y >> l), lab], 1).astype(np.int32)
                    templates[k] = ((max_x - min_x) >> l, (max_y - min_y) >> l, l, fbegin, len(f))
                    feats.append(f)
                    fbegin += len(f)
            meta.append({"verts": verts - np.array([min_x, min_y], np.float64), "plane_label": plane_label,
                         "width": max_x - min_x, "height": max_y - min_y})
        bank.classes.append((cid, templates, np.concatenate(feats, 0)))
        bank.meta[cid] = meta
    return bank


def _fill_convex(mask_shape, verts):
    """Boolean mask of the convex polygon `verts` (Kx2, x right / y down) over an HxW grid; returns (mask, bbox)."""
    H, W = mask_shape
    x0 = max(0, int(math.floor(verts[:, 0].min())))
    x1 = min(W, int(math.ceil(verts[:, 0].max())) + 1)
    y0 = max(0, int(math.floor(verts[:, 1].min())))
    y1 = min(H, int(math.ceil(verts[:, 1].max())) + 1)
    if x1 <= x0 or y1 <= y0:
        return None, (0, 0, 0, 0)
    ys, xs = np.mgrid[y0:y1, x0:x1]
    nxt = np.roll(verts, -1, 0)
    area = 0.5 * np.sum(verts[:, 0] * nxt[:, 1] - nxt[:, 0] * verts[:, 1])
    sgn = 1.0 if area > 0 else -1.0
    inside = np.ones(ys.shape, bool)
    for a, b in zip(verts, nxt):
        cross = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
        inside &= (cross * sgn) >= 0
    return inside, (y0, y1, x0, x1)


def _smooth_noise(rng, H, W, sigma_px, amp):
    """Cheap band-limited noise: upsampled coarse noise (bilinear), amplitude `amp`."""
    gh, gw = H // sigma_px + 2, W // sigma_px + 2
    coarse = rng.normal(0, amp, (gh, gw))
    yi = np.arange(H) / sigma_px
    xi = np.arange(W) / sigma_px
    y0 = yi.astype(int)
    x0 = xi.astype(int)
    fy = (yi - y0)[:, None]
    fx = (xi - x0)[None, :]
    c00 = coarse[y0][:, x0]
    c01 = coarse[y0][:, x0 + 1]
    c10 = coarse[y0 + 1][:, x0]
    c11 = coarse[y0 + 1][:, x0 + 1]
    return (c00 * (1 - fy) * (1 - fx) + c01 * (1 - fy) * fx + c10 * fy * (1 - fx) + c11 * fy * fx)


def make_scene(bank, width=640, height=480, seed=0, n_instances=4, n_distractors=6, depth=True, row_pad=0, texture=0.6):
    """-> (sources list, truth list).  sources[0] = BGR u8 (H, W, 3) view with row stride (W+row_pad)*3 bytes
    (the reference hands match() a strided ROI view: linemod_ensenso_detect_3_mult_detect_service.cpp:324-326),
    sources[1] = depth u16 mm (only if `depth` and the bank has a DepthNormal modality).
    `texture` scales the background's low-frequency noise: 0.6 leaves a gradient label on ~19 % of the level-0 pixels
    (~30 % at level 1), 1.0 on ~42 % (a very busy scene); bench.py reports the measured densities."""
    rng = np.random.default_rng([seed, 1000])
    H, W = height, width
    base = rng.uniform(140, 200, 3)
    img = np.empty((H, W, 3), np.float64)
    tex = _smooth_noise(rng, H, W, 8, 8.0 * texture)
    for c in range(3):
        img[:, :, c] = base[c] + tex + _smooth_noise(rng, H, W, 16, 5.0 * texture)
    dimg = None
    want_depth = depth and any(m["type"] == "DepthNormal" for m in bank.modalities)
    if want_depth:
        ys, xs = np.mgrid[0:H, 0:W]
        bg_phi = rng.uniform(0, 2 * np.pi)
        dimg = rng.uniform(700, 900) + 0.35 * (math.cos(bg_phi) * (xs - W / 2) + math.sin(bg_phi) * (ys - H / 2))
    truth = []
    # distractors: random convex shapes that are not in the bank
    for _ in range(n_distractors):
        v = _convex_base_shape(rng) * rng.uniform(20, 70)
        v = v + np.array([rng.uniform(40, W - 40), rng.uniform(40, H - 40)])
        m, (y0, y1, x0, x1) = _fill_convex((H, W), v)
        if m is None:
            continue
        col = base + rng.uniform(-70, 40, 3)
        for c in range(3):
            img[y0:y1, x0:x1, c][m] = col[c]
    cids = [c for c, _, _ in bank.classes]
    for k in range(n_instances):
        cid = cids[int(rng.integers(0, len(cids)))]
        meta = bank.meta[cid]
        tid = int(rng.integers(0, len(meta)))
        mt = meta[tid]
        if mt["width"] + 100 >= W or mt["height"] + 100 >= H:
            continue
        px = int(rng.integers(45, W - mt["width"] - 45))
        py = int(rng.integers(45, H - mt["height"] - 45))
        v = mt["verts"] + np.array([px, py], np.float64)
        m, (y0, y1, x0, x1) = _fill_convex((H, W), v)
        if m is None:
            continue
        col = np.clip(base - rng.uniform(85, 120, 3), 5, 255)
        for c in range(3):
            img[y0:y1, x0:x1, c][m] = col[c] + tex[y0:y1, x0:x1][m] * 0.3
        if want_depth:
            phi = mt["plane_label"] * (math.pi / 4)
            ys, xs = np.mgrid[y0:y1, x0:x1]
            cx, cy = v[:, 0].mean(), v[:, 1].mean()
            dobj = dimg[y0:y1, x0:x1].min() - 25.0 + 0.9 * (math.cos(phi) * (xs - cx) + math.sin(phi) * (ys - cy))
            dimg[y0:y1, x0:x1][m] = dobj[m]
        truth.append({"class_id": cid, "template_id": tid, "x": px, "y": py})
    noise = rng.normal(0, 1.5, (H, W, 3))
    buf = np.zeros((H, W + row_pad, 3), np.uint8)
    buf[:, :W] = np.clip(np.rint(img + noise), 0, 255).astype(np.uint8)
    dview = None
    if want_depth:
        d16 = np.clip(np.rint(dimg + rng.normal(0, 0.3, (H, W))), 1, 65535).astype(np.uint16)
        holes = rng.uniform(0, 1, (H, W)) < 0.01
        d16[holes] = 0
        dbuf = np.zeros((H, W + row_pad), np.uint16)
        dbuf[:, :W] = d16
        dview = dbuf[:, :W]
    # one source per modality, in the bank's modality order (what Detector::match asserts)
    sources = [buf[:, :W] if m["type"] == "ColorGradient" else dview for m in bank.modalities]
    if not depth:
        sources = [s for s in sources if s is not None]
    return sources, truth
