"""Block-level model of k_score_coarse_sb's table (build_device_bank: features -> 8 shift classes -> same-shift triples emitted round
robin -> blocks of 5 triples, leftovers padded) and of what a per-frame BLOCK order would buy:
  table      today's arrival order (modalities interleaved 3:3), blocks in table order
  sorted     arrival order = DepthNormal features by label, then ColorGradient features; blocks in table order
  perframe   the `sorted` table, blocks processed in ascending order of their expected response in THIS frame
             (sum over the block's real features of the frame's mean response of (modality, label), divided by their number)
Every processed block costs 15 loads per live chunk, padded entries included, as in the kernel.
usage: python scripts/sim_score_blocks.py synth|mesh [threshold] [texture]"""
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


def blocks_of(arrival):
    """arrival: list of (m, label, e0) -> list of blocks, each a list of up to 15 entries (None = padding)"""
    cls = [[] for _ in range(8)]
    for ft in arrival:
        cls[ft[2] & 7].append(ft)
    taken = [0] * 8
    groups = []
    any_ = True
    while any_:
        any_ = False
        for k in range(8):
            if len(cls[k]) - taken[k] >= 3:
                groups.append(cls[k][taken[k]:taken[k] + 3])
                taken[k] += 3
                any_ = True
    for k in range(8):
        if taken[k] < len(cls[k]):
            rest = cls[k][taken[k]:]
            groups.append(rest + [None] * (3 - len(rest)))
    return [sum(groups[i:i + 5], []) for i in range(0, len(groups), 5)]


def loads_of(blocks, order, flat, pos, raw_thr, nf):
    rows = []
    cnt = []
    for b in order:
        real = [ft for ft in blocks[b] if ft is not None]
        rows.append(sum(flat[m][l][e:e + pos] for (m, l, e) in real) if real else np.zeros(pos, np.int32))
        cnt.append(len(real))
    loads = 0
    for c0 in range(0, pos, 504):
        S = np.zeros(min(504, pos - c0), np.int32)
        seen = 0
        for r, c in zip(rows, cnt):
            S = S + r[c0:c0 + 504]
            seen += c
            loads += 15
            if not (S >= raw_thr + 1 - 4 * (nf - seen)).any():
                break
    return loads


tot = {"table": 0, "sorted": 0, "perframe": 0}
nblk = {"table": 0, "sorted": 0}
n = 0
for fr in frames:
    od.match(fr, thr)
    lm = [od.linear_memory(1, m, (H // 2, W // 2)).astype(np.int32) for m in range(M)]
    flat = [np.concatenate([x.reshape(8, -1), np.zeros((8, cells + 64), np.int32)], 1) for x in lm]
    mean_resp = [[float(flat[m][l][:T * T * cells].mean()) for l in range(8)] for m in range(M)]
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
        inter = []
        i = [0, 0]
        while i[0] < len(feats[0]) or i[1] < len(feats[1]):
            for m in range(M):
                for _ in range(3):
                    if i[m] < len(feats[m]):
                        inter.append(feats[m][i[m]])
                        i[m] += 1
        nf = len(inter)
        raw_thr = int(2 * nf + thr / 100 * 2 * nf + 0.5)
        b_tab = blocks_of(inter)
        b_sorted = blocks_of(sorted(feats[1], key=lambda ft: ft[1]) + feats[0])
        tot["table"] += loads_of(b_tab, range(len(b_tab)), flat, pos, raw_thr, nf)
        tot["sorted"] += loads_of(b_sorted, range(len(b_sorted)), flat, pos, raw_thr, nf)
        keys = []
        for b in b_sorted:
            real = [ft for ft in b if ft is not None]
            keys.append(sum(mean_resp[m][l] for (m, l, e) in real) / max(1, len(real)))
        tot["perframe"] += loads_of(b_sorted, np.argsort(keys, kind="stable"), flat, pos, raw_thr, nf)
        nblk["table"] += len(b_tab)
        nblk["sorted"] += len(b_sorted)
        n += 1
print(kind, "thr", thr, "tex", tex, "templates", n, " loads per wave:", "  ".join("%s %.1f" % (k, v / n) for k, v in tot.items()),
      " blocks per row: table %.2f sorted %.2f" % (nblk["table"] / n, nblk["sorted"] / n))
