"""Template-bank container shared by the host bindings, the synthetic generator and the tests.

Mirrors the data held by cv::linemod::Detector: `T_at_level`, the modality list and, per
class id, a vector of TemplatePyramids, each `pyramid_levels * n_modalities` Templates indexed `l*M + m`
(reference readers: /root/reference/src/rgbdDetector.cpp:1668-1680; writers: src/renderer.cpp:56-70).

Flat array form (what crosses the C ABI, include/lmx.h `lmx_bank_add_class`):
  templates int32 [n_pyramids * L * M, 5] = (width, height, pyramid_level, feat_begin, feat_count)
  features  int32 [total_features, 3]      = (x, y, label)
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

DEFAULT_COLOR_GRADIENT = {"type": "ColorGradient", "weak_threshold": 10.0, "num_features": 63, "strong_threshold": 55.0}
DEFAULT_DEPTH_NORMAL = {"type": "DepthNormal", "distance_threshold": 2000, "difference_threshold": 50,
                        "num_features": 63, "extract_threshold": 2}


@dataclass
class TemplateBank:
    T: List[int]
    modalities: List[Dict]
    classes: List[Tuple[str, np.ndarray, np.ndarray]] = field(default_factory=list)
    # optional generator side-car (not part of the wire format): per class, per pyramid, shape info
    meta: Dict = field(default_factory=dict)
    # DepthNormal's NORMAL_LUT[20][20][20] (upstream normal_lut.i: one-hot labels, indexed [v3][v2][v1]); None = the library's
    # default table (lmx_default_normal_lut).  Carried to the device bank by NativeBank.from_bank and to the oracle.
    normal_lut: Optional[np.ndarray] = None

    @property
    def pyramid_levels(self):
        return len(self.T)

    @property
    def n_modalities(self):
        return len(self.modalities)

    def num_templates(self, class_id=None):
        per = self.pyramid_levels * self.n_modalities
        return sum(t.shape[0] // per for cid, t, _ in self.classes if class_id is None or cid == class_id)

    def class_ids(self):
        return sorted(cid for cid, _, _ in self.classes)

    def get_templates(self, class_id, template_id):
        """-> list of (width, height, pyramid_level, features[n,3]) of length L*M (cv::linemod::Detector::getTemplates)."""
        per = self.pyramid_levels * self.n_modalities
        for cid, t, f in self.classes:
            if cid == class_id:
                out = []
                for k in range(per):
                    w, h, lvl, b, n = t[template_id * per + k]
                    out.append((int(w), int(h), int(lvl), f[b:b + n].copy()))
                return out
        raise KeyError(class_id)

    def shard(self, rank, world):
        """Contiguous template_id ranges per rank and class: -> {class_id: (begin, end)}."""
        per = self.pyramid_levels * self.n_modalities
        out = {}
        for cid, t, _ in self.classes:
            n = t.shape[0] // per
            out[cid] = ((rank * n) // world, ((rank + 1) * n) // world)
        return out
