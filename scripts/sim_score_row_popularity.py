"""would ordering each template's FIRST feature block by bank-wide row popularity make co-resident waves of a CU hit the same
lines in the vector L1?  A "row" is one (modality, orientation, y mod T, x mod T) linear memory of the coarsest level (600 bytes nibble-packed at
640x480); k_score_coarse_sb reads, per feature, 252 consecutive bytes of one row per 504-placement chunk.  Feature order inside a template is free
(sums commute), so the first scalar block could prefer the rows most templates of the bank use.
Model (CPU, numpy, oracle linear memories): one CU of the XCD that scores frame 0 -- it gets every 32nd workgroup of 4 consecutive templates; 32
resident waves issue one 15-feature block per turn, round robin, a finished wave is replaced by the next template; exact pruning per chunk as in the
kernel (scripts/sim_score_order.py's rule); the vector L1 is a 32 KB LRU of 128-byte lines.  Reported per order: wave loads per wave (pruning
depth), L1 line accesses, L1 misses = L2 read requests.
  current     the table's order (modalities interleaved in groups of 3, the template's own feature order)
  popularity  features sorted by how many features of the WHOLE bank lie on the same row, most popular first
  pop-first15 only the first block takes the 15 most popular rows, the rest keeps the current order
usage: python scripts/sim_score_row_popularity.py synth|mesh [threshold]"""
import collections
import sys

sys.path.insert(0, '/root/repo')
import numpy as np  # noqa: E402
from linemod_pose_estimation_amd import synth  # noqa: E402
from oracle import oracle as o  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "synth"
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 92.0
W, H, T = 640, 480, 8
if kind == "mesh":
    from linemod_pose_estimation_amd import meshsynth as ms
    bank, _, _, _ = ms.load_bank("memoryChip2")
    chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
    frame = ms.make_scene(chip, views, seed=7000, n_instances=3, other_tri=cpu, n_other=2, texture=0.6)[0]
else:
    bank = synth.make_bank(3000, seed=20250215)
    frame = synth.make_scene(bank, W, H, seed=3000, texture=0.6)[0]
od = o.OracleDetector(bank)
cid, tarr, farr = bank.classes[0]
L, M = 2, 2
Wc, Hc = W // 2 // T, H // 2 // T
cells = Wc * Hc
ORI_STRIDE = ((T * T * cells + 1) // 2 + cells // 2 + 2048 + 64 + 255) // 256 * 256     # LevelGeom::nib_ori_stride
MOD_STRIDE = 8 * ORI_STRIDE + 8192
od.match(frame, thr)
flat = [np.concatenate([od.linear_memory(1, m, (H // 2, W // 2)).astype(np.int32).reshape(8, -1), np.zeros((8, cells + 64), np.int32)], 1) for m in range(M)]
N = bank.num_templates()

# every template's coarse features as (m, label, e0, row id) in the table's order; bank-wide popularity of the rows
feats, pop = [], collections.Counter()
for t in range(N):
    per_m = []
    for m in range(M):
        w, h, lv, fb, fc = tarr[(t * L + 1) * M + m]
        f = farr[fb:fb + fc]
        grid = (f[:, 1] % T) * T + (f[:, 0] % T)
        e0 = grid * cells + (f[:, 1] // T) * Wc + (f[:, 0] // T)
        per_m.append([(m, int(l), int(e), (m, int(l), int(g))) for l, e, g in zip(f[:, 2], e0, grid)])
    order, i = [], [0, 0]
    while i[0] < len(per_m[0]) or i[1] < len(per_m[1]):
        for m in range(M):
            for _ in range(3):
                if i[m] < len(per_m[m]):
                    order.append(per_m[m][i[m]])
                    i[m] += 1
    feats.append(order)
    for x in order:
        pop[x[3]] += 1
rows_sorted = sorted(pop.values(), reverse=True)
print("%s bank: %d templates, %d features, %d distinct rows of %d possible; the 53 most popular rows (= 32 KB of L1) hold %.1f %% of all features, the top 15 rows %.1f %%"
      % (kind, N, sum(rows_sorted), len(rows_sorted), M * 8 * T * T, 100.0 * sum(rows_sorted[:53]) / sum(rows_sorted), 100.0 * sum(rows_sorted[:15]) / sum(rows_sorted)))


def positions(t):
    w, h = tarr[(t * L + 1) * M][0], tarr[(t * L + 1) * M][1]
    wf, hf = (w - 1) // T + 1, (h - 1) // T + 1
    return max(0, min((Hc - hf) * Wc + (Wc - wf) + 1, cells))


def wave_program(t, order):
    """-> list of turns; a turn = the line numbers one block of the template touches (all live chunks), following the kernel's pruning."""
    pos = positions(t)
    nf = len(order)
    raw_thr = int(2 * nf + thr / 100 * 2 * nf + 0.5)
    turns = collections.defaultdict(list)
    for c0 in range(0, pos, 504):
        n = min(504, pos - c0)
        S = np.zeros(n, np.int32)
        for bi, b in enumerate(range(0, nf, 15)):
            for (m, l, e, _) in order[b:b + 15]:
                S += flat[m][l][e + c0:e + c0 + n]
                addr = m * 64 * MOD_STRIDE + l * ORI_STRIDE + ((e + c0) >> 3) * 4       # 64 frames per lane: modality blocks are 64 frames apart
                turns[bi].extend(range(addr // 128, (addr + 255) // 128 + 1))
            rem = nf - min(nf, b + 15)
            if not (S >= raw_thr + 1 - 4 * rem).any():
                break
    return [turns[k] for k in sorted(turns)]


def simulate(order_of):
    mine = [t for t in range(N) if (t // 4) % 32 == 0]          # this CU's workgroups: every 32nd group of 4 templates
    progs = collections.deque(wave_program(t, order_of(feats[t])) for t in mine)
    loads = sum(len(turn) for p in progs for turn in p)
    n_waves = len(progs)
    lru = collections.OrderedDict()
    acc = miss = 0
    live = [progs.popleft() for _ in range(min(32, len(progs)))]
    while live:
        nxt = []
        for p in live:
            for line in p.pop(0):
                acc += 1
                if line in lru:
                    lru.move_to_end(line)
                else:
                    miss += 1
                    lru[line] = True
                    if len(lru) > 256:
                        lru.popitem(last=False)
            if p:
                nxt.append(p)
            elif progs:
                nxt.append(progs.popleft())
        live = nxt
    return n_waves, loads, acc, miss


def by_popularity(order):
    return sorted(order, key=lambda x: -pop[x[3]])


def first15(order):
    s = by_popularity(order)[:15]
    rest = [x for x in order if x not in s]
    return s + rest


base = None
for name, fn in (("current", lambda o_: o_), ("popularity", by_popularity), ("pop-first15", first15)):
    n_waves, loads, acc, miss = simulate(fn)
    lines_per_wave = acc / n_waves
    if base is None:
        base = miss
    print("%-12s waves %4d   line accesses per wave %7.1f   L1 hit rate %5.1f %%   L2 requests per wave %7.1f  (%+.1f %% vs current)"
          % (name, n_waves, lines_per_wave, 100.0 * (acc - miss) / acc, miss / n_waves, 100.0 * (miss - base) / base))
