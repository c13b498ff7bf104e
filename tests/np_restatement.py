"""Second, independent restatement (numpy, whole-array formulations) of the stages of cv::linemod::Detector::match
from SURVEY.md Appendix A.  Test infrastructure only: it pins the C++ oracle (oracle/linemod_oracle.cpp), which
is written as per-pixel loops, against a differently structured implementation of the same published algorithm
(the reference itself has no tests or fixtures for this path: SURVEY.md section 4).
Reference call site of the restated path: /root/reference/src/rgbdDetector.cpp:31-34.
"""
import numpy as np

F32 = np.float32


# ---- A.6: the LUT from its generating rule (not from the table) ------------------------------------------
def similarity_lut_from_rule():
    lut = np.zeros(256, np.uint8)
    for ori in range(8):
        for nib in range(16):
            lo = hi = 0
            for j in range(4):
                if nib >> j & 1:
                    d = abs(ori - j)
                    lo = max(lo, max(0, 4 - min(d, 8 - d)))      # low nibble: circular distance
                    hi = max(hi, max(0, 4 - abs(ori - (j + 4))))  # high nibble: NOT circular (upstream asymmetry)
            lut[32 * ori + nib] = lo
            lut[32 * ori + 16 + nib] = hi
    return lut


# ---- A.2 ---------------------------------------------------------------------------------------------------
def gaussian7(img):
    k = np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    a = img.astype(np.int64)
    if a.ndim == 2:
        a = a[:, :, None]
    p = np.pad(a, ((0, 0), (3, 3), (0, 0)), mode="edge")
    rows = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(7))
    p = np.pad(rows, ((3, 3), (0, 0), (0, 0)), mode="edge")
    out = sum(k[i] * p[i:i + a.shape[0]] for i in range(7))
    out = np.minimum((out + (1 << 15)) >> 16, 255).astype(np.uint8)
    return out.reshape(img.shape)


def sobel3(sm):
    a = sm.astype(np.int32)
    if a.ndim == 2:
        a = a[:, :, None]
    p = np.pad(a, ((1, 1), (1, 1), (0, 0)), mode="edge")
    H, W = a.shape[:2]
    def s(dy, dx):
        return p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    gx = (s(-1, 1) + 2 * s(0, 1) + s(1, 1)) - (s(-1, -1) + 2 * s(0, -1) + s(1, -1))
    gy = (s(1, -1) + 2 * s(1, 0) + s(1, 1)) - (s(-1, -1) + 2 * s(-1, 0) + s(-1, 1))
    return gx.astype(np.int16).reshape(sm.shape), gy.astype(np.int16).reshape(sm.shape)


def fast_atan2_deg(y, x):
    """vectorised cv::fastAtan2, float32 throughout, same operation order."""
    y = y.astype(F32)
    x = x.astype(F32)
    scale = F32(180 / np.pi)
    p1 = F32(0.9997878412794807) * scale
    p3 = F32(-0.3258083974640975) * scale
    p5 = F32(0.1555786518463281) * scale
    p7 = F32(-0.04432655554792128) * scale
    eps = F32(2.220446049250313e-16)
    ax, ay = np.abs(x), np.abs(y)
    big = ax >= ay
    num = np.where(big, ay, ax)
    den = np.where(big, ax, ay) + eps
    c = (num / den).astype(F32)
    c2 = (c * c).astype(F32)
    poly = ((((p7 * c2).astype(F32) + p5).astype(F32) * c2).astype(F32) + p3).astype(F32)
    poly = (((poly * c2).astype(F32) + p1).astype(F32) * c).astype(F32)
    a = np.where(big, poly, (F32(90.0) - poly).astype(F32)).astype(F32)
    a = np.where(x < 0, (F32(180.0) - a).astype(F32), a)
    a = np.where(y < 0, (F32(360.0) - a).astype(F32), a)
    return a.astype(F32)


def quantized_orientations(bgr, weak_threshold=10.0):
    sm = gaussian7(bgr)
    dx, dy = sobel3(sm)
    dx = dx.astype(np.int32)
    dy = dy.astype(np.int32)
    mag3 = dx * dx + dy * dy
    m0, m1, m2 = mag3[..., 0], mag3[..., 1], mag3[..., 2]
    c0 = (m0 >= m1) & (m0 >= m2)
    c1 = ~c0 & (m1 >= m0) & (m1 >= m2)
    sel = np.where(c0, 0, np.where(c1, 1, 2))
    ii, jj = np.indices(sel.shape)
    sx, sy, mag = dx[ii, jj, sel], dy[ii, jj, sel], mag3[ii, jj, sel].astype(F32)
    ang = fast_atan2_deg(sy, sx)
    q = np.clip(np.rint((ang * F32(16.0 / 360.0)).astype(F32)), 0, 255).astype(np.uint8)  # rint = half to even
    q[0, :] = 0
    q[-1, :] = 0
    q[:, 0] = 0
    q[:, -1] = 0
    q[1:-1, 1:-1] &= 7
    H, W = q.shape
    votes = np.zeros((8, H, W), np.int32)
    lab = q & 7
    for dy_ in (-1, 0, 1):
        for dx_ in (-1, 0, 1):
            sh = np.zeros((H, W), np.int64) - 1
            ys = slice(max(0, -dy_), H - max(0, dy_))
            xs = slice(max(0, -dx_), W - max(0, dx_))
            yd = slice(max(0, dy_), H - max(0, -dy_))
            xd = slice(max(0, dx_), W - max(0, -dx_))
            sh[ys, xs] = lab[yd, xd]
            for b in range(8):
                votes[b] += (sh == b)
    best = votes.argmax(0)            # first maximum, like upstream's strict '<' scan
    maxv = votes.max(0)
    out = np.zeros((H, W), np.uint8)
    ok = (mag > F32(weak_threshold) * F32(weak_threshold)) & (maxv >= 5)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    out[ok] = (1 << best[ok]).astype(np.uint8)
    return out, mag


# ---- A.3 ---------------------------------------------------------------------------------------------------
def pyrdown(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    a = img.astype(np.int64)
    if a.ndim == 2:
        a = a[:, :, None]
    H, W = a.shape[:2]
    p = np.pad(a, ((0, 0), (2, 2), (0, 0)), mode="reflect")
    rows = sum(k[i] * p[:, i:i + W:2][:, :W // 2] for i in range(5))
    p = np.pad(rows, ((2, 2), (0, 0), (0, 0)), mode="reflect")
    out = sum(k[i] * p[i:i + H:2][:H // 2] for i in range(5))
    out = ((out + 128) >> 8).astype(np.uint8)
    return out.reshape((H // 2, W // 2) + img.shape[2:])


# ---- A.4 (with the restatement-defined NORMAL_LUT rule of DESIGN.md) ------------------------------------------
def normal_label(v2, v1):
    cx = 2 * v1 - 19
    cy = 2 * v2 - 19
    a, b = np.abs(cx), np.abs(cy)
    horiz = 2 * a * b < a * a - b * b
    vert = ~horiz & (2 * a * b < b * b - a * a)
    k = np.where(horiz, np.where(cx > 0, 0, 4),
                 np.where(vert, np.where(cy > 0, 2, 6),
                          np.where(cx > 0, np.where(cy > 0, 1, 7), np.where(cy > 0, 3, 5))))
    return (1 << k).astype(np.uint8)


def median5(img):
    p = np.pad(img, 2, mode="edge")
    H, W = img.shape
    st = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(5) for dx in range(5)], 0)
    return np.sort(st, 0)[12]


def default_normal_lut():
    """The restatement-defined default NORMAL_LUT[20][20][20]: azimuth sector of the cell centre, nz ignored."""
    v2, v1 = np.indices((20, 20))
    return np.broadcast_to(normal_label(v2, v1), (20, 20, 20)).copy()


def quantized_normals(depth, distance_threshold=2000, difference_threshold=50, normal_lut=None):
    d = depth.astype(np.int64)
    H, W = d.shape
    r = 5
    out = np.zeros((H, W), np.uint8)
    ys, xs = slice(r, H - r - 1), slice(r, W - r - 1)
    c = d[ys, xs]
    A0 = np.zeros_like(c); A1 = np.zeros_like(c); A3 = np.zeros_like(c); b0 = np.zeros_like(c); b1 = np.zeros_like(c)
    for j in (-r, 0, r):
        for i in (-r, 0, r):
            if i == 0 and j == 0:
                continue
            nb = d[r + j:H - r - 1 + j, r + i:W - r - 1 + i]
            delta = nb - c
            f = (np.abs(delta) < difference_threshold).astype(np.int64)
            A0 += f * i * i; A1 += f * i * j; A3 += f * j * j
            b0 += f * i * delta; b1 += f * j * delta
    det = A0 * A3 - A1 * A1
    ddx = A3 * b0 - A1 * b1
    ddy = -A1 * b0 + A0 * b1
    nx = (1150 * ddx).astype(F32)
    ny = (1150 * ddy).astype(F32)
    nz = (-det * c).astype(F32)
    s = np.sqrt(((nx * nx).astype(F32) + (ny * ny).astype(F32)).astype(F32) + (nz * nz).astype(F32)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1.0) / s).astype(F32)
        v1 = ((nx * inv).astype(F32) * F32(10) + F32(10)).astype(F32)
        v2 = ((ny * inv).astype(F32) * F32(10) + F32(10)).astype(F32)
        v3 = ((nz * inv).astype(F32) * F32(20) + F32(20)).astype(F32)
    good = (c < distance_threshold) & (s > 0)
    v1i = np.where(good, v1, 0).astype(np.int64)   # truncation toward zero, values are >= 0
    v2i = np.where(good, v2, 0).astype(np.int64)
    v3i = np.where(good, v3, 0).astype(np.int64)
    # NORMAL_LUT[v3][v2][v1] with C's flat layout; indices past the table (upstream UB) give no label
    lut = (default_normal_lut() if normal_lut is None else np.asarray(normal_lut, np.uint8)).reshape(8000)
    idx = (v3i * 20 + v2i) * 20 + v1i
    lab = np.where(idx < 8000, lut[np.minimum(idx, 7999)], 0).astype(np.uint8)
    out[ys, xs] = np.where(good, lab, 0)
    return median5(out), out


# ---- A.5 - A.7 -----------------------------------------------------------------------------------------------
def spread(q, T):
    H, W = q.shape
    out = np.zeros_like(q)
    for r in range(T):
        for c in range(T):
            out[:H - r, :W - c] |= q[r:, c:]
    return out


def response_maps(spr, lut=None):
    lut = similarity_lut_from_rule() if lut is None else lut
    lo, hi = spr & 15, spr >> 4
    return np.stack([np.maximum(lut[32 * o + lo], lut[32 * o + 16 + hi]) for o in range(8)], 0)


def linearize(rmap, T):
    H, W = rmap.shape
    return np.stack([rmap[r0::T, c0::T].reshape(-1) for r0 in range(T) for c0 in range(T)], 0)


# ---- A.8 - A.10 (python loops: small cases only) ----------------------------------------------------------------
def _lm_read(lm_o, base, n):
    """flat read of `n` elements starting at `base` of one orientation's [T*T, cells] matrix; zero past the end."""
    flat = lm_o.reshape(-1)
    out = np.zeros(n, np.int64)
    hi = min(len(flat), base + n)
    if hi > base:
        out[:hi - base] = flat[base:hi]
    return out


def similarity(lm, size_wh, T, templ_wh, feats):
    w, h = size_wh
    Wc, Hc = w // T, h // T
    wf, hf = (templ_wh[0] - 1) // T + 1, (templ_wh[1] - 1) // T + 1
    positions = (Hc - hf) * Wc + (Wc - wf) + 1
    dst = np.zeros(Wc * Hc, np.int64)
    for x, y, label in feats:
        if x < 0 or x >= w or y < 0 or y >= h:
            continue
        base = ((y % T) * T + x % T) * (Wc * Hc) + (y // T) * Wc + x // T
        if positions > 0:
            dst[:positions] += _lm_read(lm[label], base, positions)
    return (dst & 255).astype(np.uint8).reshape(Hc, Wc), positions


def similarity_local(lm, size_wh, T, feats, cx, cy):
    w, h = size_wh
    Wc, Hc = w // T, h // T
    ox, oy = (int(cx / T) - 8) * T, (int(cy / T) - 8) * T   # int(): C++ truncation toward zero
    dst = np.zeros((16, 16), np.int64)
    for x, y, label in feats:
        x, y = x + ox, y + oy
        if x < 0 or y < 0 or x >= w or y >= h:
            continue
        base = ((y % T) * T + x % T) * (Wc * Hc) + (y // T) * Wc + x // T
        for r in range(16):
            dst[r] += _lm_read(lm[label], base + r * Wc, 16)
    return (dst & 255).astype(np.uint8)


def raw_threshold(nf, thr):
    return int(F32(2 * nf) + (F32(thr) / F32(100.0)) * F32(2 * nf) + F32(0.5))


def match(bank, sources, threshold):
    """Whole Detector::match for a TemplateBank; returns pre-sort matches as tuples
    (x, y, similarity(float32), class_id, template_id) in upstream insertion order."""
    L, M = len(bank.T), len(bank.modalities)
    lms, sizes = [], []
    color = [None] * M
    quant = [None] * M
    H, W = sources[0].shape[:2]
    for l in range(L):
        T = bank.T[l]
        if l > 0:
            H, W = H // 2, W // 2
        lv = []
        for m, mod in enumerate(bank.modalities):
            if mod["type"] == "ColorGradient":
                color[m] = np.ascontiguousarray(sources[m]) if l == 0 else pyrdown(color[m])
                q, _ = quantized_orientations(color[m], mod["weak_threshold"])
            else:
                q = quantized_normals(np.ascontiguousarray(sources[m]), mod["distance_threshold"], mod["difference_threshold"],
                                      getattr(bank, "normal_lut", None))[0] \
                    if l == 0 else quant[m][::2, ::2][:H, :W].copy()
            quant[m] = q
            r = response_maps(spread(q, T))
            lv.append(np.stack([linearize(r[o], T) for o in range(8)], 0))
        lms.append(lv)
        sizes.append((W, H))
    out = []
    per = L * M
    for cid, templates, feats in sorted(bank.classes, key=lambda c: c[0]):
        for tid in range(templates.shape[0] // per):
            tp = [templates[tid * per + k] for k in range(per)]
            fl = [[tuple(int(v) for v in f) for f in feats[t[3]:t[3] + t[4]]] for t in tp]
            Tl = bank.T[-1]
            w, h = sizes[-1]
            tot = np.zeros((h // Tl, w // Tl), np.int64)
            nf = 0
            for m in range(M):
                k = (L - 1) * M + m
                s, _ = similarity(lms[-1][m], sizes[-1], Tl, (tp[k][0], tp[k][1]), fl[k])
                tot += s
                nf += len(fl[k])
            rt = raw_threshold(nf, threshold)
            cands = []
            off = Tl // 2 + (Tl % 2 - 1)
            for r in range(tot.shape[0]):
                for c in range(tot.shape[1]):
                    if tot[r, c] > rt:
                        sim = F32(F32(int(tot[r, c])) * F32(100.0) / F32(4 * nf)) + F32(0.5)
                        cands.append([c * Tl + off, r * Tl + off, F32(sim)])
            for l in range(L - 2, -1, -1):
                T = bank.T[l]
                w, h = sizes[l]
                border, off = 8 * T, T // 2 + (T % 2 - 1)
                max_x, max_y = w - tp[l * M][0] - border, h - tp[l * M][1] - border
                keep = []
                for x, y, _ in cands:
                    x, y = 2 * x + 1, 2 * y + 1
                    x, y = max(x, border), max(y, border)
                    x, y = min(x, max_x), min(y, max_y)
                    tot2 = np.zeros((16, 16), np.int64)
                    nf2 = 0
                    for m in range(M):
                        k = l * M + m
                        tot2 += similarity_local(lms[l][m], sizes[l], T, fl[k], x, y)
                        nf2 += len(fl[k])
                    best, br, bc = 0, -1, -1
                    for r in range(16):
                        for c in range(16):
                            if tot2[r, c] > best:
                                best, br, bc = int(tot2[r, c]), r, c
                    nx = (int(x / T) - 8 + bc) * T + off
                    ny = (int(y / T) - 8 + br) * T + off
                    sim = F32(F32(best) * F32(100.0) / F32(4 * nf2))
                    if not sim < F32(threshold):
                        keep.append([nx, ny, sim])
                cands = keep
            out += [(x, y, F32(s), cid, tid) for x, y, s in cands]
    return out
