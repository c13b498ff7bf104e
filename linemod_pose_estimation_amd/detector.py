"""Python host-side mirror of the reference's operator surface for the matching path.

`Detector` keeps the names and argument meaning of cv::linemod::Detector as the reference uses it:
  readLinemod(filename)                    /root/reference/src/rgbdDetector.cpp:1668-1680
  detector.match(sources, threshold, ...)  /root/reference/src/rgbdDetector.cpp:31-34
  detector.classIds(), getTemplates(), numTemplates(), getT(), pyramidLevels()
Everything computes through the C ABI of liblmx.so (include/lmx.h) on a gfx950 device; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib
from .bank import TemplateBank

MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"), ("class_index", "<i4")])
RAW_MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                            ("class_index", "<i4"), ("frame", "<i4"), ("order_key", "<u8")])


def _images(frames):
    """frames: list (per frame) of list (per modality) of numpy arrays -> ctypes Image array (+ keep-alive)."""
    flat = [s for fr in frames for s in fr]
    arr = (_lib.Image * len(flat))()
    for i, s in enumerate(flat):
        if s.dtype == np.uint8 and s.ndim == 3:
            ch, es = s.shape[2], 1
            ok = s.strides[2] == 1 and s.strides[1] == ch
        elif s.dtype == np.uint8 and s.ndim == 2:        # MONO8 (upload_raw with mono=True)
            ch, es = 1, 1
            ok = s.strides[1] == 1
        elif s.dtype == np.uint16 and s.ndim == 2:
            ch, es = 1, 2
            ok = s.strides[1] == 2
        elif s.dtype == np.float32 and s.ndim == 2:      # depth in metres (upload_raw with depth_float_m=True)
            ch, es = 1, 4
            ok = s.strides[1] == 4
        else:
            raise TypeError("sources must be uint8 HxWx3 / HxW (colour) or uint16 / float32 HxW (depth)")
        if not ok:
            raise TypeError("source pixels must be contiguous within a row (row stride may be larger)")
        arr[i] = _lib.Image(s.ctypes.data, s.shape[0], s.shape[1], ch, es, s.strides[0])
    return arr, flat


class PreparedBatch:
    """lmx_image descriptors of a batch of host frames, built once (the arrays are kept alive by this object)."""

    def __init__(self, frames):
        self.imgs, self.keep = _images(frames)
        self.n_frames, self.n_sources = len(frames), len(frames[0])

    def slice(self, first, count):
        """Frames [first, first + count) of this batch as a PreparedBatch of their own (shares the descriptors and the arrays)."""
        out = PreparedBatch.__new__(PreparedBatch)
        out.imgs = (_lib.Image * (count * self.n_sources)).from_address(C.addressof(self.imgs) + first * self.n_sources * C.sizeof(_lib.Image))
        out.keep = (self.imgs, self.keep)
        out.n_frames, out.n_sources = count, self.n_sources
        return out


class NativeBank:
    """Owns an lmx_bank handle (host template state of cv::linemod::Detector)."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    @classmethod
    def from_bank(cls, bank: TemplateBank):
        L = _lib.lib()
        T = (C.c_int32 * len(bank.T))(*bank.T)
        mods = (_lib.ModalityDesc * len(bank.modalities))()
        for i, m in enumerate(bank.modalities):
            if m["type"] == "ColorGradient":
                mods[i] = _lib.ModalityDesc(_lib.LMX_MOD_COLOR_GRADIENT, m.get("weak_threshold", 10.0), m.get("strong_threshold", 55.0),
                                            m.get("num_features", 63), 0, 0, 0)
            elif m["type"] == "DepthNormal":
                mods[i] = _lib.ModalityDesc(_lib.LMX_MOD_DEPTH_NORMAL, 0.0, 0.0, m.get("num_features", 63),
                                            m.get("distance_threshold", 2000), m.get("difference_threshold", 50),
                                            m.get("extract_threshold", 2))
            else:
                raise ValueError("unknown modality %r" % (m["type"],))
        desc = _lib.BankDesc(len(bank.T), T, len(bank.modalities), mods)
        h = C.c_void_p()
        _lib.check(L.lmx_bank_create(C.byref(desc), C.byref(h)))
        self = cls(h.value)
        per = len(bank.T) * len(bank.modalities)
        for cid, templates, features in bank.classes:
            templates = np.ascontiguousarray(templates, np.int32)
            features = np.ascontiguousarray(features, np.int32)
            _lib.check(L.lmx_bank_add_class(self.h, cid.encode(), templates.shape[0] // per,
                                            templates.ctypes.data_as(C.POINTER(C.c_int32)),
                                            features.ctypes.data_as(C.POINTER(C.c_int32)), features.shape[0]))
        if getattr(bank, "normal_lut", None) is not None:
            self.set_normal_lut(bank.normal_lut)
        return self

    def set_normal_lut(self, lut):
        """DepthNormal's NORMAL_LUT[20][20][20] (upstream normal_lut.i); None = choose the default generator explicitly."""
        if lut is None:
            _lib.check(_lib.lib().lmx_bank_set_normal_lut(self.h, None))
        else:
            lut = np.ascontiguousarray(lut, np.uint8).reshape(_lib.LMX_NORMAL_LUT_SIZE)
            _lib.check(_lib.lib().lmx_bank_set_normal_lut(self.h, lut.ctypes.data))

    def load_normal_lut(self, path):
        _lib.check(_lib.lib().lmx_bank_load_normal_lut(self.h, str(path).encode()))

    def normal_lut(self):
        out = np.empty((20, 20, 20), np.uint8)
        _lib.check(_lib.lib().lmx_bank_get_normal_lut(self.h, out.ctypes.data))
        return out

    def normal_lut_origin(self):
        return int(_lib.lib().lmx_bank_normal_lut_origin(self.h))

    @classmethod
    def create(cls, T, modalities):
        """Empty bank: cv::linemod::Detector(modalities, T) as in the reference's trainers (src/renderer.cpp:179-185)."""
        return cls.from_bank(TemplateBank(T=list(T), modalities=list(modalities)))

    def add_template(self, sources, class_id, object_mask=None, device=0):
        """cv::linemod::Detector::addTemplate (reference src/renderer.cpp:308): -> (template_id or -1, bounding box (x, y, w, h))."""
        imgs, keep = _images([list(sources)])
        mask_img = None
        if object_mask is not None:
            m = object_mask
            if m.dtype != np.uint8 or m.ndim != 2 or m.strides[1] != 1:
                raise TypeError("object_mask must be uint8 HxW")
            mask_img = _lib.Image(m.ctypes.data, m.shape[0], m.shape[1], 1, 1, m.strides[0])
        tid = C.c_int32(-1)
        bb = (C.c_int32 * 4)()
        _lib.check(_lib.lib().lmx_bank_add_template(self.h, device, imgs, len(sources), class_id.encode(),
                                                    C.byref(mask_img) if mask_img is not None else None, C.byref(tid), bb))
        del keep
        return tid.value, tuple(bb)

    @classmethod
    def load_yaml(cls, path):
        h = C.c_void_p()
        _lib.check(_lib.lib().lmx_bank_load_yaml(str(path).encode(), C.byref(h)))
        return cls(h.value)

    def save_yaml(self, path):
        _lib.check(_lib.lib().lmx_bank_save_yaml(self.h, str(path).encode()))

    def to_bank(self) -> TemplateBank:
        L = _lib.lib()
        nl, nm = L.lmx_bank_pyramid_levels(self.h), L.lmx_bank_num_modalities(self.h)
        T = [L.lmx_bank_T(self.h, l) for l in range(nl)]
        mods = []
        for i in range(nm):
            d = _lib.ModalityDesc()
            _lib.check(L.lmx_bank_modality(self.h, i, C.byref(d)))
            if d.type == _lib.LMX_MOD_COLOR_GRADIENT:
                mods.append({"type": "ColorGradient", "weak_threshold": d.weak_threshold, "num_features": d.num_features,
                             "strong_threshold": d.strong_threshold})
            else:
                mods.append({"type": "DepthNormal", "distance_threshold": d.distance_threshold,
                             "difference_threshold": d.difference_threshold, "num_features": d.num_features,
                             "extract_threshold": d.extract_threshold})
        bank = TemplateBank(T=T, modalities=mods)
        if self.normal_lut_origin() in (_lib.LMX_LUT_USER, _lib.LMX_LUT_SIDECAR):
            bank.normal_lut = self.normal_lut()
        per = nl * nm
        for ci in range(L.lmx_bank_num_classes(self.h)):
            cid = L.lmx_bank_class_id(self.h, ci)
            n = L.lmx_bank_num_templates(self.h, cid)
            templates = np.zeros((n * per, 5), np.int32)
            feats = []
            fb = 0
            w, h, lv, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            fp = C.POINTER(C.c_int32)()
            for t in range(n):
                for k in range(per):
                    _lib.check(L.lmx_bank_get_template(self.h, cid, t, k, C.byref(w), C.byref(h), C.byref(lv), C.byref(fp), C.byref(nf)))
                    templates[t * per + k] = (w.value, h.value, lv.value, fb, nf.value)
                    if nf.value:
                        feats.append(np.ctypeslib.as_array(fp, shape=(nf.value, 3)).copy())
                    fb += nf.value
            bank.classes.append((cid.decode(), templates, np.concatenate(feats, 0) if feats else np.zeros((0, 3), np.int32)))
        return bank

    def class_ids(self):
        L = _lib.lib()
        return [L.lmx_bank_class_id(self.h, i).decode() for i in range(L.lmx_bank_num_classes(self.h))]

    def __del__(self):
        try:
            if self.h:
                _lib.lib().lmx_bank_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Detector:
    """Device-resident detector: a template bank (or one rank's shard of it) in HBM plus per-frame workspaces.

    Mirrors cv::linemod::Detector for the matching side.  `match` is the drop-in for the call in
    rgbdDetector::linemod_detection (/root/reference/src/rgbdDetector.cpp:33)."""

    def __init__(self, bank, width, height, device=0, max_batch=1, max_candidates=0, shard_rank=0, shard_world=1, stream=None, hipgraph=False, overlap=False,
                 async_input=False):
        if isinstance(bank, TemplateBank):
            self.native_bank = NativeBank.from_bank(bank)
            self.bank = bank
        elif isinstance(bank, NativeBank):
            self.native_bank = bank
            self.bank = bank.to_bank()
        else:
            raise TypeError("bank must be a TemplateBank or NativeBank")
        self.width, self.height, self.max_batch = width, height, max_batch
        desc = _lib.CtxDesc(device, width, height, max_batch, max_candidates, shard_rank, shard_world, stream, (1 if hipgraph else 0) | (2 if overlap else 0) | (4 if async_input else 0))
        self.h = C.c_void_p()
        _lib.check(_lib.lib().lmx_ctx_create(self.native_bank.h, C.byref(desc), C.byref(self.h)))
        self._class_ids = self.native_bank.class_ids()
        self.max_outstanding = int(_lib.lib().lmx_ctx_max_outstanding(self.h))   # enqueues that may be in flight before a collect

    # ---- cv::linemod::Detector-style accessors -----------------------------------------------------------
    @classmethod
    def readLinemod(cls, filename, width, height, **kw):
        return cls(NativeBank.load_yaml(filename), width, height, **kw)

    def classIds(self):
        return list(self._class_ids)

    def numTemplates(self, class_id=None):
        return _lib.lib().lmx_bank_num_templates(self.native_bank.h, class_id.encode() if class_id else None)

    def getTemplates(self, class_id, template_id):
        return self.bank.get_templates(class_id, template_id)

    def getT(self, level):
        return self.bank.T[level]

    def pyramidLevels(self):
        return self.bank.pyramid_levels

    # ---- matching -------------------------------------------------------------------------------------------
    @staticmethod
    def _cids(class_ids):
        n = len(class_ids)
        arr = (C.c_char_p * max(1, n))(*[c.encode() for c in class_ids])
        return arr, n

    def match(self, sources, threshold, class_ids=(), cap=1 << 14):
        """One frame through `lmx_match`, the drop-in for Detector::match.  Returns MATCH_DTYPE records in upstream
        output order (std::sort + std::unique applied)."""
        L = _lib.lib()
        imgs, keep = _images([sources])
        cids, ncid = self._cids(class_ids)
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_size_t()
        _lib.check(L.lmx_match(self.h, imgs, len(sources), C.c_float(threshold), cids, ncid, out.ctypes.data, cap, C.byref(n)))
        del keep
        return out[:n.value].copy()

    def match_masked(self, sources, masks, threshold, class_ids=(), cap=1 << 14):
        """Detector::match(sources, threshold, matches, class_ids, noArray(), masks): masks = one uint8 HxW array per modality (None
        entries = no mask for that source)."""
        imgs, keep = _images([sources])
        marr = (_lib.Image * len(sources))()
        for i, m in enumerate(masks):
            if m is None:
                marr[i] = _lib.Image(None, 0, 0, 1, 1, 0)
            else:
                if m.dtype != np.uint8 or m.ndim != 2 or m.strides[1] != 1:
                    raise TypeError("masks must be uint8 HxW")
                marr[i] = _lib.Image(m.ctypes.data, m.shape[0], m.shape[1], 1, 1, m.strides[0])
        cids, ncid = self._cids(class_ids)
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_size_t()
        _lib.check(_lib.lib().lmx_match_masked(self.h, imgs, marr, len(sources), C.c_float(threshold), cids, ncid, out.ctypes.data, cap, C.byref(n)))
        del keep
        return out[:n.value].copy()

    def upload_masks(self, masks):
        """masks: list (per frame of the most recent upload) of list (per modality) of uint8 HxW arrays or None."""
        flat = [m for fr in masks for m in fr]
        marr = (_lib.Image * len(flat))()
        for i, m in enumerate(flat):
            marr[i] = _lib.Image(None, 0, 0, 1, 1, 0) if m is None else _lib.Image(m.ctypes.data, m.shape[0], m.shape[1], 1, 1, m.strides[0])
        _lib.check(_lib.lib().lmx_ctx_upload_masks(self.h, len(masks), marr, len(masks[0])))

    def match_prepared(self, batch, threshold, cap=1 << 12):
        """`lmx_match` on a one-frame PreparedBatch: the lmx_image descriptors were built once (what a C++ caller holding cv::Mat
        headers passes), so the call costs what the library costs, not the marshalling."""
        if getattr(self, "_mp_out", None) is None or len(self._mp_out) < cap:
            self._mp_out = np.zeros(cap, MATCH_DTYPE)
            self._mp_n = C.c_size_t()
            self._mp_cids = (C.c_char_p * 1)()
        _lib.check(_lib.lib().lmx_match(self.h, batch.imgs, batch.n_sources, C.c_float(threshold), self._mp_cids, 0, self._mp_out.ctypes.data, cap, C.byref(self._mp_n)))
        return self._mp_out[:self._mp_n.value].copy()

    def match_batch(self, frames, threshold, class_ids=(), cap=1 << 12):
        """n frames through `lmx_match_batch` (per-frame output capacity `cap`)."""
        L = _lib.lib()
        imgs, keep = _images(frames)
        cids, ncid = self._cids(class_ids)
        out = np.zeros((len(frames), cap), MATCH_DTYPE)
        n_out = (C.c_size_t * len(frames))()
        _lib.check(L.lmx_match_batch(self.h, len(frames), imgs, len(frames[0]), C.c_float(threshold), cids, ncid,
                                     out.ctypes.data, cap, n_out))
        del keep
        return [out[f, :n_out[f]].copy() for f in range(len(frames))]

    # split-phase API
    def upload(self, frames):
        """frames: list (per frame) of list (per modality) of numpy arrays, or a batch prepared once with `prepare_batch` (a caller
        that uploads the same host buffers repeatedly -- a camera ring, the bench -- skips rebuilding the lmx_image descriptors)."""
        if isinstance(frames, PreparedBatch):
            _lib.check(_lib.lib().lmx_ctx_upload(self.h, frames.n_frames, frames.imgs, frames.n_sources))
            return
        imgs, keep = _images(frames)
        _lib.check(_lib.lib().lmx_ctx_upload(self.h, len(frames), imgs, len(frames[0])))
        del keep

    @staticmethod
    def prepare_batch(frames):
        return PreparedBatch(frames)

    def upload_wait(self):
        """Host-side wait for the most recent upload's transfer (needed only with async_input and pinned sources)."""
        _lib.check(_lib.lib().lmx_ctx_upload_wait(self.h))

    def upload_raw(self, frames, src_size, crop_xy=(0, 0), blur3=True, mono=False, depth_float_m=False):
        """Raw camera frames + the node-side steps in front of match() on the device (lmx_ctx_upload_raw): optional
        MONO8->BGR, GaussianBlur 3x3 on the full frame, crop to the context size, float-metre depth -> u16 mm."""
        pre = _lib.PreDesc(src_size[0], src_size[1], crop_xy[0], crop_xy[1], int(blur3), int(mono), int(depth_float_m))
        if isinstance(frames, PreparedBatch):   # descriptors built once (a camera ring, the bench)
            _lib.check(_lib.lib().lmx_ctx_upload_raw(self.h, frames.n_frames, frames.imgs, frames.n_sources, C.byref(pre)))
            return
        imgs, keep = _images(frames)
        _lib.check(_lib.lib().lmx_ctx_upload_raw(self.h, len(frames), imgs, len(frames[0]), C.byref(pre)))
        del keep

    def enqueue(self, n_frames, threshold, class_ids=()):
        cids, ncid = self._cids(class_ids)
        _lib.check(_lib.lib().lmx_ctx_enqueue(self.h, n_frames, C.c_float(threshold), cids, ncid))

    def collect(self, n_frames, cap_total=1 << 16):
        """-> list (per frame) of MATCH_DTYPE arrays in upstream output order (views into a reused buffer are copied)."""
        if getattr(self, "_flat", None) is None or len(self._flat) < cap_total:
            self._flat = np.zeros(cap_total, MATCH_DTYPE)
        offs = (C.c_size_t * (n_frames + 1))()
        _lib.check(_lib.lib().lmx_ctx_collect_flat(self.h, n_frames, self._flat.ctypes.data, len(self._flat), offs))
        return [self._flat[offs[f]:offs[f + 1]].copy() for f in range(n_frames)]

    def set_cluster_sidecar(self, obj_origin_dists, rects, vote_row_col_step, renderer_radius_min, renderer_radius_step, cluster_size_thresh=2):
        """Side-car of the device-side consumer chain (lmx_ctx_set_cluster_sidecar): per-template origin distances and rects."""
        d = np.ascontiguousarray(obj_origin_dists, np.float64)
        r = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
        pp = _lib.ClusterParams(int(vote_row_col_step), float(renderer_radius_min), float(renderer_radius_step), int(cluster_size_thresh))
        _lib.check(_lib.lib().lmx_ctx_set_cluster_sidecar(self.h, d.ctypes.data, r.ctypes.data, len(d), C.byref(pp)))

    def collect_clusters(self, n_frames, cap_total=1 << 16):
        """Device-side std::sort + std::unique + rcd_voting/filter/scoring/IoU-NMS on the oldest outstanding enqueue
        (lmx_ctx_collect_clusters) -> list per frame of (matches, clusters, members)."""
        m = np.zeros(cap_total, MATCH_DTYPE)
        cl = np.zeros(cap_total, CLUSTER_DTYPE)
        mem = np.zeros(cap_total, np.int32)
        mo = (C.c_size_t * (n_frames + 1))()
        co = (C.c_size_t * (n_frames + 1))()
        _lib.check(_lib.lib().lmx_ctx_collect_clusters(self.h, n_frames, m.ctypes.data, cap_total, mo, cl.ctypes.data, cap_total, co, mem.ctypes.data, cap_total))
        out = []
        for f in range(n_frames):
            c = cl[co[f]:co[f + 1]].copy()
            out.append((m[mo[f]:mo[f + 1]].copy(), c, mem))
        return out

    def raw_matches_ptrs(self):
        rec, cnt, cap = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().lmx_ctx_raw_matches(self.h, C.byref(rec), C.byref(cnt), C.byref(cap)))
        return rec.value, cnt.value, cap.value

    def export_raw(self, d_block_ptr, capacity_records):
        """One D2D copy of this context's gather block (64-byte header + capacity_records x 32 B) on the context's stream."""
        _lib.check(_lib.lib().lmx_ctx_export_raw(self.h, d_block_ptr, capacity_records))

    def export_raw_on(self, d_block_ptr, capacity_records, stream):
        """The same copy on `stream` (a raw hipStream_t), which first waits on the device for the most recent enqueue."""
        _lib.check(_lib.lib().lmx_ctx_export_raw_on(self.h, d_block_ptr, capacity_records, stream))

    def release(self):
        """Drop the oldest outstanding enqueue without a read-back (its records were consumed on the device)."""
        _lib.check(_lib.lib().lmx_ctx_release(self.h))

    def sync(self):
        _lib.check(_lib.lib().lmx_ctx_sync(self.h))

    # ---- introspection ---------------------------------------------------------------------------------------
    def level_shape(self, level):
        return self.height >> level, self.width >> level

    def debug_quantized(self, frame, level, modality):
        H, W = self.level_shape(level)
        out = np.empty((H, W), np.uint8)
        _lib.check(_lib.lib().lmx_ctx_debug_read(self.h, frame, _lib.LMX_DBG_QUANTIZED, level, modality, out.ctypes.data, out.nbytes))
        return out

    def debug_linear_memory(self, frame, level, modality):
        H, W = self.level_shape(level)
        T = self.bank.T[level]
        out = np.empty((8, T * T, (H // T) * (W // T)), np.uint8)
        _lib.check(_lib.lib().lmx_ctx_debug_read(self.h, frame, _lib.LMX_DBG_LINEAR_MEMORY, level, modality, out.ctypes.data, out.nbytes))
        return out

    def debug_depth(self, frame, modality):
        out = np.empty((self.height, self.width), np.uint16)
        _lib.check(_lib.lib().lmx_ctx_debug_read(self.h, frame, 3, 0, modality, out.ctypes.data, out.nbytes))
        return out

    def debug_pyramid_bgr(self, frame, level, modality=0):
        H, W = self.level_shape(level)
        out = np.empty((H, W, 3), np.uint8)
        _lib.check(_lib.lib().lmx_ctx_debug_read(self.h, frame, _lib.LMX_DBG_PYRAMID_BGR, level, modality, out.ctypes.data, out.nbytes))
        return out

    def stats(self):
        a, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().lmx_ctx_stats(self.h, C.byref(a), C.byref(b)))
        return {"candidates": a.value, "raw_matches": b.value}

    def set_profiling(self, on=True):
        """on: True = every kernel, False = none, or a kernel name / list of names (only those get HIP events)."""
        L = _lib.lib()
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            names = [on] if isinstance(on, str) else list(on)
            ids = {L.lmx_kernel_name(k).decode(): k for k in range(L.lmx_num_kernels())}
            mask = 0
            for n in names:
                mask |= 1 << ids[n]
        _lib.check(L.lmx_ctx_set_profiling(self.h, mask))

    def reset_profiling(self):
        _lib.check(_lib.lib().lmx_ctx_reset_profiling(self.h))

    def kernel_times(self):
        L = _lib.lib()
        out = {}
        for k in range(L.lmx_num_kernels()):
            ms, n = C.c_double(), C.c_int64()
            _lib.check(L.lmx_ctx_kernel_time(self.h, k, C.byref(ms), C.byref(n)))
            out[L.lmx_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def device_kernel_name(self, kernel_name):
        """Name of the device kernel a profiler shows for this context's `kernel_name` (e.g. k_score_coarse -> k_score_coarse_u8)."""
        L = _lib.lib()
        for k in range(L.lmx_num_kernels()):
            if L.lmx_kernel_name(k).decode() == kernel_name:
                return L.lmx_ctx_device_kernel_name(self.h, k).decode()
        raise KeyError(kernel_name)

    def algorithmic_bytes(self, kernel_name, n_frames):
        L = _lib.lib()
        for k in range(L.lmx_num_kernels()):
            if L.lmx_kernel_name(k).decode() == kernel_name:
                v = C.c_double()
                _lib.check(L.lmx_ctx_algorithmic_bytes(self.h, k, n_frames, C.byref(v)))
                return v.value
        raise KeyError(kernel_name)

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().lmx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArena:
    """Pinned host memory (lmx_host_alloc == hipHostMalloc) handed out as numpy arrays: frames placed here take the zero-copy
    upload path (DMA straight from the caller's buffer, no staging copy)."""

    def __init__(self, nbytes):
        self.p = C.c_void_p()
        _lib.check(_lib.lib().lmx_host_alloc(nbytes, C.byref(self.p)))
        self.nbytes, self.used = nbytes, 0
        self._buf = (C.c_uint8 * nbytes).from_address(self.p.value)

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        off = (self.used + 255) & ~255
        if off + n > self.nbytes:
            raise MemoryError("PinnedArena exhausted")
        self.used = off + n
        return np.frombuffer(self._buf, dtype=dtype, count=int(np.prod(shape)), offset=off).reshape(shape)

    def put(self, a):
        out = self.empty(a.shape, a.dtype)
        out[...] = a
        return out

    def close(self):
        if getattr(self, "p", None) and self.p.value:
            self._buf = None
            _lib.lib().lmx_host_free(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


CLUSTER_DTYPE = np.dtype([("index", "<i4", (3,)), ("rect", "<i4", (4,)), ("score", "<f8"), ("member_begin", "<i4"), ("member_count", "<i4")],
                         align=True)


def cluster_matches(matches, obj_origin_dists, rects, vote_row_col_step, renderer_radius_min, renderer_radius_step, cluster_size_thresh=2):
    """rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU on one frame's matches
    (reference: src/linemod_ensenso_detect_3_mult_detect_service.cpp:376-447).  Returns (clusters, members): clusters is a
    CLUSTER_DTYPE array in upstream's final order, members holds indices into `matches`."""
    matches = np.ascontiguousarray(matches, MATCH_DTYPE)
    dists = np.ascontiguousarray(obj_origin_dists, np.float64)
    rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
    pp = _lib.ClusterParams(int(vote_row_col_step), float(renderer_radius_min), float(renderer_radius_step), int(cluster_size_thresh))
    clusters = np.zeros(max(1, len(matches)), CLUSTER_DTYPE)
    members = np.zeros(max(1, len(matches)), np.int32)
    n = C.c_size_t()
    _lib.check(_lib.lib().lmx_cluster_matches(matches.ctypes.data, len(matches), dists.ctypes.data, rects.ctypes.data, len(dists), C.byref(pp),
                                              clusters.ctypes.data, len(clusters), C.byref(n), members.ctypes.data, len(members)))
    return clusters[:n.value].copy(), members


GATHER_HEADER_BYTES = 64


def merge_gathered(blocks, n_ranks, block_stride, capacity_records, n_frames, cap_total=1 << 16, frame_groups=1):
    """Host merge (C) of gathered per-rank blocks -> list (per frame) of final matches in upstream output order.  frame_groups > 1: the ranks
    form a frame_groups x template_shards grid (rank k = group k // R, shard k % R) and a rank's records carry frame indices local to its group."""
    blocks = np.ascontiguousarray(blocks, np.uint8)
    out = np.zeros(cap_total, MATCH_DTYPE)
    offs = (C.c_size_t * (n_frames + 1))()
    _lib.check(_lib.lib().lmx_merge_gathered_groups(blocks.ctypes.data, n_ranks, block_stride, capacity_records, n_frames, frame_groups,
                                                    out.ctypes.data, cap_total, offs))
    return [out[offs[f]:offs[f + 1]].copy() for f in range(n_frames)]


def merge_raw(records, cap=1 << 16):
    """Host merge of gathered raw records of ONE frame -> final matches (std::sort + std::unique, upstream order)."""
    records = np.ascontiguousarray(records, RAW_MATCH_DTYPE)
    out = np.zeros(cap, MATCH_DTYPE)
    n = C.c_size_t()
    _lib.check(_lib.lib().lmx_merge_raw(records.ctypes.data, len(records), out.ctypes.data, cap, C.byref(n)))
    return out[:n.value].copy()


def linemod_detection(detector, sources, threshold):
    """Same shape as rgbdDetector::linemod_detection (/root/reference/src/rgbdDetector.cpp:31-34): forwards to
    match(sources, threshold, matches, class_ids={}, no quantized-image output)."""
    return detector.match(sources, threshold, ())
