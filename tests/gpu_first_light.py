"""First-light check on the GPU box (a script, not collected by pytest): stage-by-stage comparison of liblmx against the oracle.
Lives under tests/ because only tests, smoke() and the bench baseline may touch oracle/."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from linemod_pose_estimation_amd import synth, Detector
from oracle import oracle as o

def cmp(name, a, b):
    ok = np.array_equal(a, b)
    print("%-28s %s" % (name, "OK" if ok else "MISMATCH %d / %d" % ((a != b).sum(), a.size)), flush=True)
    if not ok:
        idx = np.argwhere(a != b)[:8]
        for i in idx:
            print("    at", tuple(i), "gpu", a[tuple(i)], "oracle", b[tuple(i)])
    return ok

for (W, H, N, M, thr) in [(160, 80, 8, 2, 70.0), (640, 480, 300, 2, 85.0), (640, 480, 300, 1, 88.0)]:
    mods = ("ColorGradient", "DepthNormal")[:M]
    size_range = (20.0, 40.0) if W < 320 else (55.0, 194.0)
    bank = synth.make_bank(N, modalities=mods, seed=5, size_range=size_range)
    src, truth = synth.make_scene(bank, W, H, seed=7, row_pad=16)
    od = o.OracleDetector(bank)
    t = time.time(); ref = od.match(src, thr); t_cpu = time.time() - t
    det = Detector(bank, W, H)
    t = time.time(); got = det.match(src, thr); t_gpu = time.time() - t
    t = time.time(); got = det.match(src, thr); t_gpu2 = time.time() - t
    print("== %dx%d N=%d M=%d thr=%.0f: oracle %d matches (%d cands) %.1f ms | gpu %d matches %.1f/%.1f ms %s" %
          (W, H, N, M, thr, len(ref), od.last_candidates(), t_cpu * 1e3, len(got), t_gpu * 1e3, t_gpu2 * 1e3, det.stats()), flush=True)
    allok = True
    for l in range(2):
        Hl, Wl = H >> l, W >> l
        for m in range(M):
            allok &= cmp("quantized l%d m%d" % (l, m), det.debug_quantized(0, l, m), od.quantized(l, m, (Hl, Wl)))
            allok &= cmp("linear memory l%d m%d" % (l, m), det.debug_linear_memory(0, l, m), od.linear_memory(l, m, (Hl, Wl)))
    if M >= 1:
        allok &= cmp("pyrDown bgr l1", det.debug_pyramid_bgr(0, 1, 0), o.pyrdown(np.ascontiguousarray(src[0])))
    same = len(ref) == len(got) and all(np.array_equal(ref[k], got[k]) for k in ref.dtype.names)
    print("matches identical:", same, flush=True)
    if not same:
        print("ref", ref[:10]); print("got", got[:10])
    det.close()
print("done")
