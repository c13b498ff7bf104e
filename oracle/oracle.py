"""ctypes wrapper around oracle/liblinemod_oracle.so (CPU restatement; test infrastructure only).

Mirrors the stages of cv::linemod::Detector::match as restated in linemod_oracle.cpp (SURVEY.md
Appendix A); the reference call site is /root/reference/src/rgbdDetector.cpp:31-34.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblinemod_oracle.so")

MOD_COLOR_GRADIENT = 0
MOD_DEPTH_NORMAL = 1


def build(force=False):
    src = os.path.join(_HERE, "linemod_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


class MatchT(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("similarity", C.c_float),
                ("template_id", C.c_int32), ("class_index", C.c_int32)]


MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"),
                        ("template_id", "<i4"), ("class_index", "<i4")])

RAW_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                      ("class_index", "<i4"), ("frame", "<i4"), ("order_key", "<u8")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
        L.lmo_similarity_lut.restype = u8p
        L.lmo_fast_atan2.restype = C.c_float
        L.lmo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.lmo_raw_threshold.restype = C.c_int
        L.lmo_raw_threshold.argtypes = [C.c_int, C.c_float]
        L.lmo_detector_create.restype = C.c_void_p
        L.lmo_detector_create.argtypes = [C.c_int, i32p, C.c_int, f32p]
        L.lmo_detector_destroy.argtypes = [C.c_void_p]
        L.lmo_detector_set_normal_lut.argtypes = [C.c_void_p, C.c_void_p]
        L.lmo_detector_set_normal_lut.restype = None
        L.lmo_quantized_normals_lut.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lmo_quantized_normals_lut.restype = None
        L.lmo_detector_add_class.restype = C.c_int
        L.lmo_detector_add_class.argtypes = [C.c_void_p, C.c_char_p, C.c_int, i32p, i32p]
        L.lmo_detector_match.restype = C.c_long
        L.lmo_detector_match.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), i32p, i32p, C.POINTER(C.c_size_t),
                                         C.c_int, C.c_float, C.POINTER(C.c_char_p), C.c_int]
        L.lmo_detector_get_matches.restype = C.c_long
        L.lmo_detector_get_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.lmo_detector_get_raw.restype = C.c_long
        L.lmo_detector_get_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.lmo_detector_last_candidates.restype = C.c_long
        L.lmo_detector_last_candidates.argtypes = [C.c_void_p]
        L.lmo_detector_num_classes.argtypes = [C.c_void_p]
        L.lmo_detector_class_name.restype = C.c_char_p
        L.lmo_detector_class_name.argtypes = [C.c_void_p, C.c_int]
        L.lmo_detector_get_quantized.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.lmo_detector_get_linear_memory.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- stage functions ------------------------------------------------------------------------------
def similarity_lut():
    return np.ctypeslib.as_array(lib().lmo_similarity_lut(), shape=(256,)).copy()


def normal_lut():
    out = np.zeros((20, 20, 20), np.uint8)
    lib().lmo_normal_lut(_p(out))
    return out


def fast_atan2(y, x):
    return float(lib().lmo_fast_atan2(C.c_float(y), C.c_float(x)))


def pre_color(src, crop_xy, size_wh, blur3=True):
    """(MONO8->BGR) + GaussianBlur 3x3 on the full frame + crop, like the reference's detect_cb.  src: u8 HxW or HxWx3."""
    src = np.ascontiguousarray(src, np.uint8)
    SH, SW = src.shape[:2]
    SC = 1 if src.ndim == 2 else src.shape[2]
    W, H = size_wh
    out = np.empty((H, W, 3), np.uint8)
    lib().lmo_pre_color(_p(src), SH, SW, SC, C.c_size_t(SW * SC), int(crop_xy[0]), int(crop_xy[1]), H, W, int(bool(blur3)), _p(out))
    return out


def pre_depth(src_m, crop_xy, size_wh):
    """float metres -> u16 millimetres (convertTo(CV_16UC1, 1000.0)) + crop."""
    src_m = np.ascontiguousarray(src_m, np.float32)
    W, H = size_wh
    out = np.empty((H, W), np.uint16)
    lib().lmo_pre_depth(_p(src_m), C.c_size_t(src_m.shape[1]), int(crop_xy[0]), int(crop_xy[1]), H, W, _p(out))
    return out


CLUSTER_DTYPE = np.dtype([("index", "<i4", (3,)), ("rect", "<i4", (4,)), ("score", "<f8"), ("member_begin", "<i4"), ("member_count", "<i4")],
                         align=True)


def cluster_matches(matches, obj_origin_dists, rects, vote_row_col_step, renderer_radius_min, renderer_radius_step, thresh=2):
    """The reference's rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU chain."""
    matches = np.ascontiguousarray(matches, MATCH_DTYPE)
    dists = np.ascontiguousarray(obj_origin_dists, np.float64)
    rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
    clusters = np.zeros(max(1, len(matches)), CLUSTER_DTYPE)
    members = np.zeros(max(1, len(matches)), np.int32)
    L = lib()
    L.lmo_cluster_matches.restype = C.c_long
    n = L.lmo_cluster_matches(_p(matches), C.c_long(len(matches)), _p(dists), _p(rects), C.c_int(vote_row_col_step), C.c_double(renderer_radius_min),
                              C.c_double(renderer_radius_step), C.c_int(thresh), _p(clusters), _p(members))
    return clusters[:n].copy(), members


def orientation_labels(dx, dy):
    dx = np.ascontiguousarray(dx, np.int16)
    dy = np.ascontiguousarray(dy, np.int16)
    out = np.empty(dx.shape, np.uint8)
    lib().lmo_orientation_labels(_p(dx), _p(dy), C.c_size_t(dx.size), _p(out))
    return out


def raw_threshold(nf, thr):
    return int(lib().lmo_raw_threshold(int(nf), C.c_float(thr)))


def gaussian7(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    Cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    lib().lmo_gaussian7(_p(img), H, W, Cn, C.c_size_t(W * Cn), _p(out))
    return out


def sobel3(sm):
    sm = np.ascontiguousarray(sm, np.uint8)
    H, W = sm.shape[:2]
    Cn = 1 if sm.ndim == 2 else sm.shape[2]
    dx = np.empty(sm.shape, np.int16)
    dy = np.empty(sm.shape, np.int16)
    lib().lmo_sobel3(_p(sm), H, W, Cn, _p(dx), _p(dy))
    return dx, dy


def quantized_orientations(bgr, weak_threshold=10.0):
    """-> (quantized one-hot u8, magnitude^2 f32, 16->8-bin unfiltered labels)"""
    assert bgr.dtype == np.uint8 and bgr.ndim == 3 and bgr.shape[2] == 3 and bgr.strides[2] == 1 and bgr.strides[1] == 3
    H, W = bgr.shape[:2]
    q = np.empty((H, W), np.uint8)
    mag = np.empty((H, W), np.float32)
    qu = np.empty((H, W), np.uint8)
    lib().lmo_quantized_orientations(_p(bgr), H, W, C.c_size_t(bgr.strides[0]), C.c_float(weak_threshold), _p(q), _p(mag), _p(qu))
    return q, mag, qu


def pyrdown(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    Cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((H // 2, W // 2) + img.shape[2:], np.uint8)
    lib().lmo_pyrdown(_p(img), H, W, Cn, C.c_size_t(W * Cn), _p(out))
    return out


def quantized_normals(depth, distance_threshold=2000, difference_threshold=50, normal_lut=None):
    """-> (quantized after median5, before median).  normal_lut: u8 [20][20][20] (upstream normal_lut.i) or None = default table."""
    assert depth.dtype == np.uint16 and depth.ndim == 2 and depth.strides[1] == 2
    H, W = depth.shape
    out = np.empty((H, W), np.uint8)
    pre = np.empty((H, W), np.uint8)
    lut = None if normal_lut is None else np.ascontiguousarray(normal_lut, np.uint8).reshape(8000)
    lib().lmo_quantized_normals_lut(_p(depth), H, W, C.c_size_t(depth.strides[0] // 2), int(distance_threshold),
                                    int(difference_threshold), _p(out), _p(pre), None if lut is None else _p(lut))
    return out, pre


def median5(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().lmo_median5(_p(img), img.shape[0], img.shape[1], _p(out))
    return out


def spread(q, T):
    q = np.ascontiguousarray(q, np.uint8)
    out = np.empty_like(q)
    lib().lmo_spread(_p(q), q.shape[0], q.shape[1], int(T), _p(out))
    return out


def response_maps(spr):
    spr = np.ascontiguousarray(spr, np.uint8)
    out = np.empty((8,) + spr.shape, np.uint8)
    lib().lmo_response_maps(_p(spr), spr.shape[0], spr.shape[1], _p(out))
    return out


def linearize(rmap, T):
    rmap = np.ascontiguousarray(rmap, np.uint8)
    H, W = rmap.shape
    out = np.empty((T * T, (H // T) * (W // T)), np.uint8)
    lib().lmo_linearize(_p(rmap), H, W, int(T), _p(out))
    return out


def similarity(lm, size_wh, T, templ_wh, feats):
    """lm: u8 [8][T*T][W'H'];  feats int32 [n,3] -> u8 [H', W']"""
    lm = np.ascontiguousarray(lm, np.uint8)
    feats = np.ascontiguousarray(feats, np.int32)
    w, h = size_wh
    out = np.zeros((h // T, w // T), np.uint8)
    lib().lmo_similarity(_p(lm), w, h, int(T), int(templ_wh[0]), int(templ_wh[1]), _p(feats), len(feats), _p(out))
    return out


def similarity_local(lm, size_wh, T, feats, center_xy):
    lm = np.ascontiguousarray(lm, np.uint8)
    feats = np.ascontiguousarray(feats, np.int32)
    out = np.zeros((16, 16), np.uint8)
    lib().lmo_similarity_local(_p(lm), size_wh[0], size_wh[1], int(T), _p(feats), len(feats), int(center_xy[0]),
                               int(center_xy[1]), _p(out))
    return out


# ---- detector ---------------------------------------------------------------------------------------
def _modality_desc(modalities):
    rows = []
    for m in modalities:
        if m["type"] == "ColorGradient":
            rows.append([MOD_COLOR_GRADIENT, m.get("weak_threshold", 10.0), m.get("strong_threshold", 55.0),
                         m.get("num_features", 63), 0, 0, 0])
        elif m["type"] == "DepthNormal":
            rows.append([MOD_DEPTH_NORMAL, 0, 0, m.get("num_features", 63), m.get("distance_threshold", 2000),
                         m.get("difference_threshold", 50), m.get("extract_threshold", 2)])
        else:
            raise ValueError(m["type"])
    return np.asarray(rows, np.float32)


class OracleDetector:
    """CPU restatement of cv::linemod::Detector (match side).  `bank` is a TemplateBank-like object with
    .T (list), .modalities (list of dict), .classes (list of (class_id, templates[n*L*M,5], features[nf,3]))."""

    def __init__(self, bank):
        L = lib()
        self.bank = bank
        T = np.asarray(bank.T, np.int32)
        desc = _modality_desc(bank.modalities)
        self.h = C.c_void_p(L.lmo_detector_create(len(T), T.ctypes.data_as(C.POINTER(C.c_int32)), len(bank.modalities),
                                                  desc.ctypes.data_as(C.POINTER(C.c_float))))
        per = len(T) * len(bank.modalities)
        for cid, templates, features in bank.classes:
            templates = np.ascontiguousarray(templates, np.int32)
            features = np.ascontiguousarray(features, np.int32)
            assert templates.shape[0] % per == 0
            rc = L.lmo_detector_add_class(self.h, cid.encode(), templates.shape[0] // per,
                                          templates.ctypes.data_as(C.POINTER(C.c_int32)),
                                          features.ctypes.data_as(C.POINTER(C.c_int32)))
            if rc < 0:
                raise ValueError("template with more than 63 features")
        self.n_levels = len(T)
        self.n_mod = len(bank.modalities)
        if getattr(bank, "normal_lut", None) is not None:
            self.set_normal_lut(bank.normal_lut)

    def set_normal_lut(self, lut):
        """NORMAL_LUT[20][20][20] of the DepthNormal modality (upstream normal_lut.i); None = the default table."""
        if lut is None:
            lib().lmo_detector_set_normal_lut(self.h, None)
        else:
            lut = np.ascontiguousarray(lut, np.uint8).reshape(8000)
            lib().lmo_detector_set_normal_lut(self.h, _p(lut))

    def __del__(self):
        try:
            lib().lmo_detector_destroy(self.h)
        except Exception:
            pass

    def add_template(self, sources, class_id, object_mask=None):
        """Detector::addTemplate -> (template_id or -1, bounding box (x, y, w, h))."""
        L = lib()
        n = len(sources)
        data = (C.c_void_p * n)(*[s.ctypes.data for s in sources])
        rows = np.asarray([s.shape[0] for s in sources], np.int32)
        cols = np.asarray([s.shape[1] for s in sources], np.int32)
        strides = (C.c_size_t * n)(*[s.strides[0] for s in sources])
        bb = np.zeros(4, np.int32)
        mask_p, mask_stride = None, 0
        if object_mask is not None:
            assert object_mask.dtype == np.uint8 and object_mask.strides[1] == 1
            mask_p, mask_stride = object_mask.ctypes.data_as(C.c_void_p), object_mask.strides[0]
        L.lmo_detector_add_template.restype = C.c_int
        L.lmo_detector_add_template.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                C.POINTER(C.c_size_t), C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
        tid = L.lmo_detector_add_template(self.h, data, rows.ctypes.data_as(C.POINTER(C.c_int32)), cols.ctypes.data_as(C.POINTER(C.c_int32)),
                                          strides, n, class_id.encode(), mask_p, C.c_size_t(mask_stride), _p(bb))
        return tid, tuple(int(v) for v in bb)

    def get_templates(self, class_id, template_id):
        """-> list of (width, height, pyramid_level, features[n,3]) for the L*M templates of a pyramid."""
        L = lib()
        L.lmo_detector_get_template.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        out = []
        for k in range(self.n_levels * self.n_mod):
            meta = np.zeros(3, np.int32)
            feats = np.zeros((63, 3), np.int32)
            n = L.lmo_detector_get_template(self.h, class_id.encode(), template_id, k, _p(meta), _p(feats))
            assert n >= 0
            out.append((int(meta[0]), int(meta[1]), int(meta[2]), feats[:n].copy()))
        return out

    def class_ids(self):
        n = lib().lmo_detector_num_classes(self.h)
        return [lib().lmo_detector_class_name(self.h, i).decode() for i in range(n)]

    def match(self, sources, threshold, class_ids=(), masks=None):
        """sources: list of numpy arrays (BGR u8 HxWx3 / depth u16 HxW; row stride may exceed W*elem).
        masks: None, or one uint8 HxW array (or None) per source: Detector::match's `masks` argument.
        Returns structured array MATCH_DTYPE in upstream output order."""
        L = lib()
        n = len(sources)
        self._keep = sources
        data = (C.c_void_p * n)(*[s.ctypes.data for s in sources])
        rows = np.asarray([s.shape[0] for s in sources], np.int32)
        cols = np.asarray([s.shape[1] for s in sources], np.int32)
        strides = (C.c_size_t * n)(*[s.strides[0] for s in sources])
        cids = (C.c_char_p * max(1, len(class_ids)))(*[c.encode() for c in class_ids])
        mdata = mstrides = None
        if masks is not None:
            assert len(masks) == n and all(m is None or (m.dtype == np.uint8 and m.ndim == 2 and m.strides[1] == 1) for m in masks)
            mdata = (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in masks])
            mstrides = (C.c_size_t * n)(*[0 if m is None else m.strides[0] for m in masks])
        L.lmo_detector_match_masked.restype = C.c_long
        L.lmo_detector_match_masked.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p]
        rc = L.lmo_detector_match_masked(self.h, data, rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                         cols.ctypes.data_as(C.POINTER(C.c_int32)), strides, n, C.c_float(threshold), cids,
                                         len(class_ids), mdata, mstrides)
        if rc < 0:
            raise ValueError({-1: "sources.size() != modalities.size()", -2: "source sizes differ",
                              -3: "image size not a multiple of T at some level",
                              -4: "rows*cols not a multiple of 16"}.get(rc, "error %d" % rc))
        out = np.zeros(rc, MATCH_DTYPE)
        if rc:
            L.lmo_detector_get_matches(self.h, _p(out), rc)
        return out

    def last_raw(self):
        """Matches of the last match() in upstream insertion order, before sort/unique (lmx_raw_match_t layout)."""
        n = lib().lmo_detector_get_raw(self.h, None, 0)
        out = np.zeros(n, RAW_DTYPE)
        if n:
            lib().lmo_detector_get_raw(self.h, _p(out), n)
        return out

    def last_candidates(self):
        return int(lib().lmo_detector_last_candidates(self.h))

    def quantized(self, level, modality, shape):
        out = np.empty(shape, np.uint8)
        assert lib().lmo_detector_get_quantized(self.h, level, modality, _p(out)) == 0
        return out

    def linear_memory(self, level, modality, shape_hw):
        T = self.bank.T[level]
        H, W = shape_hw
        out = np.empty((8, T * T, (H // T) * (W // T)), np.uint8)
        assert lib().lmo_detector_get_linear_memory(self.h, level, modality, _p(out)) == 0
        return out
