"""Randomised trainer sweep on the GPU: lmx_bank_add_template (device quantisation + host feature selection) against the oracle's
restatement of Detector::addTemplate, on rendered views of random size / modality set / T / feature count / mask use."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import train_util
from linemod_pose_estimation_amd import NativeBank
from linemod_pose_estimation_amd.bank import TemplateBank, DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL
from oracle import oracle as o

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
done, added, t0 = 0, 0, time.time()
while done < n_cfg:
    mods = [["ColorGradient", "DepthNormal"], ["ColorGradient"], ["DepthNormal"], ["DepthNormal", "ColorGradient"]][int(rng.integers(0, 4))]
    T = [[5, 8], [4, 8], [5], [4, 4, 8]][int(rng.integers(0, 4))]
    unit = int(np.lcm.reduce([T[l] << l for l in range(len(T))]))
    W, H = unit * int(rng.integers(max(2, 160 // unit), max(3, 400 // unit) + 1)), unit * int(rng.integers(max(2, 120 // unit), max(3, 320 // unit) + 1))
    if any(((W >> l) * (H >> l)) % 16 for l in range(len(T))) or W > 480 or H > 400:
        continue
    nfeat = int(rng.choice([16, 40, 63]))
    mdesc = []
    for m in mods:
        d = dict(DEFAULT_COLOR_GRADIENT) if m == "ColorGradient" else dict(DEFAULT_DEPTH_NORMAL)
        d["num_features"] = nfeat
        mdesc.append(d)
    od = o.OracleDetector(TemplateBank(T=T, modalities=mdesc))
    nb = NativeBank.create(T, mdesc)
    what = dict(W=W, H=H, T=T, mods=mods, nfeat=nfeat)
    n_ok = 0
    for k in range(4):
        seed = int(rng.integers(0, 1 << 30))
        v = train_util.rendered_view(seed, W, H, size_range=(min(W, H) * 0.25, min(W, H) * 0.6))
        if v is None:
            continue
        bgr, depth, mask = v
        src = [bgr if m == "ColorGradient" else depth for m in mods]
        use_mask = mask if rng.integers(0, 4) else None
        ref_tid, ref_bb = od.add_template(src, "obj", use_mask)
        got_tid, got_bb = nb.add_template(src, "obj", use_mask)
        assert got_tid == ref_tid and (ref_tid < 0 or got_bb == ref_bb), (what, seed, got_tid, ref_tid, got_bb, ref_bb)
        n_ok += ref_tid >= 0
    if n_ok:
        trained = nb.to_bank()
        assert trained.num_templates("obj") == n_ok
        for tid in range(n_ok):
            for (w, h, lvl, f), (rw, rh, rl, rf) in zip(trained.get_templates("obj", tid), od.get_templates("obj", tid)):
                assert (w, h, lvl) == (rw, rh, rl) and np.array_equal(f, rf), (what, tid)
    added += n_ok
    done += 1
print("trainer fuzz ok: %d configurations, %d templates added and compared, %.0f s" % (done, added, time.time() - t0))
