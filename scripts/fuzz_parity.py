"""Randomised parity sweep on the GPU: image sizes, pyramid depths, T, modality sets, feature counts, thresholds, batch sizes and
row strides drawn at random (seeded), every configuration's stages and matches compared with the oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector, _lib
from oracle import oracle as o

def same(a, b, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for k in ("x", "y", "similarity", "template_id", "class_index"):
        assert np.array_equal(a[k], b[k]), (what, k)

import os
MAX_W, MAX_H = int(os.environ.get("FUZZ_MAX_W", "480")), int(os.environ.get("FUZZ_MAX_H", "400"))   # e.g. 1280 x 960 for a few large draws
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
done, skipped, t0 = 0, 0, time.time()
while done < n_cfg:
    L = int(rng.choice([1, 2, 2, 2, 3]))
    T = [int(rng.choice([4, 5, 6, 8])) for _ in range(L)]
    mods = [("ColorGradient",), ("DepthNormal",), ("ColorGradient", "DepthNormal"), ("DepthNormal", "ColorGradient")][int(rng.integers(0, 4))]
    # sizes: every level's size must be a multiple of its T, rows * cols a multiple of 16
    unit = int(np.lcm.reduce([T[l] << l for l in range(L)]))
    W = unit * int(rng.integers(max(1, 96 // unit), max(2, (MAX_W * 5 // 6) // unit) + 1))
    H = unit * int(rng.integers(max(1, 96 // unit), max(2, (MAX_H * 4 // 5) // unit) + 1))
    if any(((W >> l) * (H >> l)) % 16 for l in range(L)) or W > MAX_W or H > MAX_H or min(W, H) < 96:
        skipped += 1
        continue
    nfeat = int(rng.choice([8, 20, 31, 40, 63]))
    ntmpl = int(rng.integers(3, 40))
    thr = float(rng.choice([45.0, 60.0, 75.0, 85.0, 92.0]))
    B = int(rng.choice([1, 1, 2, 5, 9]))
    row_pad = int(rng.choice([0, 0, 4, 24]))
    seed = int(rng.integers(0, 1 << 30))
    what = dict(W=W, H=H, T=T, mods=mods, nfeat=nfeat, ntmpl=ntmpl, thr=thr, B=B, row_pad=row_pad, seed=seed)
    classes = [None, None, ["obj", "other"], ["b", "a", "c"]][int(rng.integers(0, 4))]
    what["classes"] = classes
    kw = {} if classes is None else {"classes": classes}
    bank = synth.make_bank(ntmpl, modalities=mods, T=tuple(T), seed=seed, num_features=nfeat, size_range=(16.0, max(20.0, min(W, H) * 0.45)), **kw)
    frames = [synth.make_scene(bank, W, H, seed=seed + 1 + f, row_pad=row_pad, texture=float(rng.choice([0.3, 0.6, 1.0])))[0] for f in range(B)]
    od = o.OracleDetector(bank)
    what["overlap"], what["hipgraph"] = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    try:
        det = Detector(bank, W, H, max_batch=B, max_candidates=1 << 19, overlap=what["overlap"], hipgraph=what["hipgraph"])
    except _lib.LmxError as e:
        print("refused", what, str(e)[:100], flush=True)
        skipped += 1
        continue
    det.upload(frames)
    det.enqueue(B, thr)
    got = det.collect(B, cap_total=1 << 20)
    for f in range(B):
        ref = od.match(frames[f], thr)
        same(got[f], ref, what)
        if f == B - 1:
            for l in range(L):
                for m in range(len(mods)):
                    assert np.array_equal(det.debug_quantized(f, l, m), od.quantized(l, m, (H >> l, W >> l))), ("quant", l, m, what)
                    assert np.array_equal(det.debug_linear_memory(f, l, m), od.linear_memory(l, m, (H >> l, W >> l))), ("lm", l, m, what)
    # the synchronous single call (lmx_match: for one or two frames the host stores them straight into device memory), with a class
    # filter drawn from the bank's classes plus a name it does not have
    names = (classes or ["obj"]) + ["nope"]
    cids = tuple(str(x) for x in rng.choice(names, size=int(rng.integers(0, 3)), replace=False))
    what["cids"] = cids
    first, ref1 = det.match(frames[0], thr, cids, cap=1 << 20), od.match(frames[0], thr, cids)
    if len(first) != len(ref1):   # diagnostics before the assertion fires
        st1 = det.stats()
        again = det.match(frames[0], thr, cids, cap=1 << 20)
        print("single call: got %d, expected %d, stats %s; repeated call: %d, stats %s" % (len(first), len(ref1), st1, len(again), det.stats()), flush=True)
        det.upload([frames[0]]); det.enqueue(1, thr); print("split-phase call:", len(det.collect(1, cap_total=1 << 20)[0]), det.stats(), flush=True)
    same(first, ref1, what)
    if rng.integers(0, 3) == 0:
        # Detector::match's masks argument: blocky random masks, sometimes only for one modality
        ms_ = []
        for m in range(len(mods)):
            if len(mods) > 1 and rng.integers(0, 3) == 0:
                ms_.append(None)
                continue
            blk = (rng.uniform(0, 1, ((H + 7) // 8, (W + 7) // 8)) < 0.75).astype(np.uint8) * 255
            ms_.append(np.ascontiguousarray(np.kron(blk, np.ones((8, 8), np.uint8))[:H, :W]))
        what["masks"] = [m is not None for m in ms_]
        same(det.match_masked(frames[0], ms_, thr, cids, cap=1 << 20), od.match(frames[0], thr, cids, masks=ms_), what)
    if rng.integers(0, 4) == 0:
        # the C++ device group with its members sharing this GPU (peer-copy collective), pipelined over a few batches
        from linemod_pose_estimation_amd.dist import DeviceGroup
        members = int(rng.choice([2, 3, 5]))
        what["group"] = members
        g = DeviceGroup(bank, W, H, members, devices=[0] * members, max_batch=B, gather_capacity=int(rng.choice([8, 8192])), max_candidates=1 << 19,
                        collective="peer_copy", overlap=bool(rng.integers(0, 2)), hipgraph=bool(rng.integers(0, 2)))
        outs_g, queued = [], 0
        for b in range(3):
            if queued == g.depth:
                outs_g.append(g.finish(B, cap=1 << 17)); queued -= 1
            g.upload(frames)
            g.submit(B, thr); queued += 1
        while queued:
            outs_g.append(g.finish(B, cap=1 << 17)); queued -= 1
        for res in outs_g:
            for f in range(B):
                same(res[f], od.match(frames[f], thr), what)
        g.close()
    if mods == ("ColorGradient", "DepthNormal") and rng.integers(0, 2):
        # node-side pre-processing on the device (lmx_ctx_upload_raw): a larger raw frame, optional MONO8 -> BGR, 3x3 blur, crop,
        # float-metre depth -> u16 mm, against the oracle's restatement of the reference's detect_cb steps
        SW, SH = W + int(rng.integers(0, 5)) * 8, H + int(rng.integers(0, 3)) * 8
        cx, cy = int(rng.integers(0, SW - W + 1)), int(rng.integers(0, SH - H + 1))
        mono, blur, fdepth = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        (bgr, depth), _ = synth.make_scene(bank, SW, SH, seed=seed + 77, texture=0.8)
        color = np.ascontiguousarray(bgr[:, :, 1]) if mono else np.ascontiguousarray(bgr)
        if fdepth:
            z = depth.astype(np.float32) / np.float32(1000.0)
            z[depth == 0] = np.nan
            ref_d = o.pre_depth(z, (cx, cy), (W, H))
        else:
            z = np.ascontiguousarray(depth)
            ref_d = np.ascontiguousarray(depth[cy:cy + H, cx:cx + W])
        det.upload_raw([[color, z]], (SW, SH), (cx, cy), blur3=blur, mono=mono, depth_float_m=fdepth)
        det.enqueue(1, thr)
        got_raw = det.collect(1, cap_total=1 << 20)[0]
        what["raw"] = dict(SW=SW, SH=SH, cx=cx, cy=cy, mono=mono, blur=blur, fdepth=fdepth)
        same(got_raw, od.match([o.pre_color(color, (cx, cy), (W, H), blur), ref_d], thr), what)
    if classes is not None and rng.integers(0, 2):
        # what the multi-GPU job does, on one GPU: sharded contexts, raw records read back, host merge in any arrival order
        from linemod_pose_estimation_amd import merge_raw, RAW_MATCH_DTYPE, PinnedArena
        world = int(rng.choice([2, 3, 5]))
        recs = []
        for r in range(world):
            sd = Detector(bank, W, H, shard_rank=r, shard_world=world, max_candidates=1 << 19)
            sd.upload([frames[0]]); sd.enqueue(1, thr)
            _, _, cap = sd.raw_matches_ptrs()
            arena = PinnedArena(64 + cap * 32 + 4096)                    # the rank's gather block, exported by the library's own copy kernel
            host = arena.empty((64 + cap * 32,), np.uint8)               # straight into pinned host memory (device-visible)
            host[:64] = 0
            sd.export_raw(host.ctypes.data, cap)
            sd.sync()
            n_rec = int(host[:64].view(np.uint32)[1])
            recs.append(host[64:64 + n_rec * 32].view(RAW_MATCH_DTYPE).copy())
            del host
            sd.close()
            arena.close()
        what["world"] = world
        same(merge_raw(np.concatenate(recs[::-1])), od.match(frames[0], thr), what)
    if rng.integers(0, 2) and classes is None:
        # the consumer chain on the device (lmx_ctx_collect_clusters) against the oracle's restatement of the reference's functions
        n_t = ntmpl
        dists = 0.5 + 0.1 * (np.arange(n_t) % 6) + rng.uniform(-0.005, 0.005, n_t)
        rects = np.stack([np.zeros(n_t), np.zeros(n_t), [m["width"] for m in bank.meta["obj"]], [m["height"] for m in bank.meta["obj"]]], 1).astype(np.int32)
        step, cthr = int(rng.choice([2, 4, 8])), int(rng.choice([1, 2, 3]))
        det.set_cluster_sidecar(dists, rects, step, 0.5, 0.1, cthr)
        det.upload(frames)   # the single call above uploaded one frame: an enqueue reads the most recent upload
        det.enqueue(B, thr)
        gc = det.collect_clusters(B, cap_total=1 << 20)
        for f in range(B):
            ref_m = od.match(frames[f], thr)
            ref_c, ref_mem = o.cluster_matches(ref_m, dists, rects, step, 0.5, 0.1, cthr)
            m, c, mem = gc[f]
            same(m, ref_m, what)
            assert len(c) == len(ref_c), ("clusters", what, len(c), len(ref_c))
            for k in ("index", "rect", "score", "member_count"):
                assert np.array_equal(c[k], ref_c[k]), ("clusters", k, what)
            for a, b in zip(c, ref_c):
                assert np.array_equal(mem[a["member_begin"]:a["member_begin"] + a["member_count"]], ref_mem[b["member_begin"]:b["member_begin"] + b["member_count"]]), ("members", what)
    det.close()
    done += 1
    if done % 10 == 0:
        print("%d configurations ok (%.0f s); last: %s, %d matches in its last frame" % (done, time.time() - t0, what, len(ref)), flush=True)
print("fuzz ok: %d configurations, %d draws skipped" % (done, skipped))
