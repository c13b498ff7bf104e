"""Does the ORDER in which the coarse scoring kernel reads a template's features matter for its exact pruning?  Simulated on the oracle's
linear memories (CPU, numpy) with the kernel's own scheme (chunks of 504 placements, bound test after every block of 15 features):
  current   the table's order: modalities interleaved in groups of 3, features in the template's own order
  bestcase  per (template, frame) the features sorted by ascending mean response over that frame's placements: no static order, and no
            order chosen without reading the memories first, can prune earlier than this on average -- the upper bound of the idea
  label     features sorted by the mean response of their (modality, label) over THIS frame: 16 numbers per frame that a kernel in front of
            the scoring could produce, and that the table order would have to follow frame by frame
  batch     the same key averaged over the frames of the run: ONE order for a whole batch (what a host-side re-sort of the table could
            follow); in these scenes the supporting plane's tilt, hence DepthNormal's dominant label, is drawn per frame
usage: python scripts/sim_score_order.py synth|mesh [threshold] [texture]"""
import sys

sys.path.insert(0, '/root/repo')
import numpy as np  # noqa: E402
from linemod_pose_estimation_amd import synth  # noqa: E402
from oracle import oracle as o  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "synth"
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 92.0
tex = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
W, H, T = 640, 480, 8
if kind == "mesh":
    from linemod_pose_estimation_amd import meshsynth as ms
    bank, _, _, _ = ms.load_bank("memoryChip2")
    chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
    frames = [ms.make_scene(chip, views, seed=7000 + f, n_instances=3, other_tri=cpu, n_other=2, texture=tex)[0] for f in range(3)]
else:
    bank = synth.make_bank(3000, seed=20250215)
    frames = [synth.make_scene(bank, W, H, seed=3000 + f, texture=tex)[0] for f in range(3)]
od = o.OracleDetector(bank)
cid, tarr, farr = bank.classes[0]
L, M = 2, 2
Wc, Hc = W // 2 // T, H // 2 // T
cells = Wc * Hc
rng = np.random.default_rng(0)
tsel = rng.choice(bank.num_templates(), 300, replace=False)


def loads_of(rows, nf, raw_thr, pos):
    loads = 0
    for c0 in range(0, pos, 504):
        S = np.zeros(min(504, pos - c0), np.int32)
        for b in range(0, nf, 15):
            S += rows[b:b + 15, c0:c0 + 504].sum(0)
            loads += min(15, nf - b)
            rem = nf - min(nf, b + 15)
            if not (S >= raw_thr + 1 - 4 * rem).any():
                break
    return loads


tot = {"current": 0, "bestcase": 0, "label": 0, "batch": 0}
n = 0
flats = []
for fr in frames:
    od.match(fr, thr)
    lm = [od.linear_memory(1, m, (H // 2, W // 2)).astype(np.int32) for m in range(M)]
    flats.append([np.concatenate([x.reshape(8, -1), np.zeros((8, cells + 64), np.int32)], 1) for x in lm])
batch_resp = np.mean([[[float(fl[m][l][:T * T * cells].mean()) for l in range(8)] for m in range(M)] for fl in flats], 0)
for flat in flats:
    # mean response of label l in modality m over the whole frame
    mean_resp = [[float(flat[m][l][:T * T * cells].mean()) for l in range(8)] for m in range(M)]
    print("   mean response per label, DepthNormal:", np.round(mean_resp[1], 2), " ColorGradient:", np.round(mean_resp[0], 2))
    for t in tsel:
        feats = []
        for m in range(M):
            w, h, lv, fb, fc = tarr[(t * L + 1) * M + m]
            f = farr[fb:fb + fc]
            e0 = ((f[:, 1] % T) * T + (f[:, 0] % T)) * cells + (f[:, 1] // T) * Wc + (f[:, 0] // T)
            feats.append([(m, int(l), int(e)) for (l, e) in zip(f[:, 2], e0)])
        w, h = tarr[(t * L + 1) * M][0], tarr[(t * L + 1) * M][1]
        wf, hf = (w - 1) // T + 1, (h - 1) // T + 1
        pos = max(0, min((Hc - hf) * Wc + (Wc - wf) + 1, cells))
        if pos == 0:
            continue
        order = []
        i = [0, 0]
        while i[0] < len(feats[0]) or i[1] < len(feats[1]):
            for m in range(M):
                for _ in range(3):
                    if i[m] < len(feats[m]):
                        order.append(feats[m][i[m]])
                        i[m] += 1
        nf = len(order)
        raw_thr = int(2 * nf + thr / 100 * 2 * nf + 0.5)
        rows = np.stack([flat[m][l][e:e + pos] for (m, l, e) in order])
        tot["current"] += loads_of(rows, nf, raw_thr, pos)
        tot["bestcase"] += loads_of(rows[np.argsort(rows.mean(1), kind="stable")], nf, raw_thr, pos)
        key = np.asarray([mean_resp[m][l] for (m, l, e) in order])
        tot["label"] += loads_of(rows[np.argsort(key, kind="stable")], nf, raw_thr, pos)
        keyb = np.asarray([batch_resp[m][l] for (m, l, e) in order])
        tot["batch"] += loads_of(rows[np.argsort(keyb, kind="stable")], nf, raw_thr, pos)
        n += 1
print(kind, "thr", thr, "tex", tex, "templates", n, " loads per wave:", "  ".join("%s %.1f" % (k, v / n) for k, v in tot.items()))
