#!/usr/bin/env python3
"""bench.py -- throughput of the LINEMOD matching hot path on MI355X.

A "step" is one pass of the hot path (the replacement of cv::linemod::Detector::match,
/root/reference/src/rgbdDetector.cpp:33) over one batch of synthetic RGB-D frames: quantise -> spread -> response maps /
linear memories -> score every (template, location) -> refine -> read the match records back -> std::sort + std::unique on
the host.  Workload at N=1 = BASELINE.json configs[1]: 640x480 RGB-D, ColorGradient + DepthNormal, 3000 templates,
T = {5, 8}; 64 frames per step by default.

`value` (the contract's headline) is measured with the frames already resident in HBM when the timed region starts
(`config.input` = "device-resident"); the K timed steps are software-pipelined over the context's output slots (K enqueues,
K collects), which alternate between the device lanes (streams).  Next to it the same JSON line carries, each timed the same
way on its own context:
  host_frames         the reference's boundary: FRESH pageable host frames every step (threaded staging copy -> pinned ->
                      DMA on a copy stream, pipelined against the kernels); its PCIe rate; this is what `speedup_vs_cpu_1core`
                      is computed from
  host_frames_pinned  the same with the frames in pinned memory (DMA straight from the caller's buffers)
  extra.busy_scene    scene texture 1.0 (42 % label density instead of 18 %: pruning in the score kernel is data dependent)
  extra.low_threshold threshold 50: pruning defeated, candidate lists explode (fewer frames per step, one lane)
  roofline            the dominant kernel against the limit it actually runs into: for the scoring kernel the
                      L2 -> L1 fill bandwidth (PMC read requests x 128 B / its exclusive launch time vs 34.5 TB/s); the VALU-issue view
                      (instructions per launch vs 1024 SIMDs x 2.4 GHz / 2 cycles) and the HBM view (algorithmic bytes, measured
                      traffic) stay in the same object
  cpu_baseline        the oracle (kind "port") single-threaded: median / p10 / p90 over >= 20 individually timed frames

N > 1 (launched by torch.distributed.run, one rank per GPU): the template bank is sharded (3000 templates per rank, weak
scaling), every rank pre-processes the same frames, and per-rank raw matches are exchanged by one RCCL all-gather per step;
`value` counts frames x (total templates / 3000) per second, `frames_per_sec` is the plain frame rate.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEMPLATES_PER_GPU = 3000
WIDTH, HEIGHT = 640, 480
THRESHOLD = 92.0  # the reference's operating threshold for the memory chip (launch/start_object_detection.launch:8)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMD-32, max shader clock (same guide)
VALU_PEAK_GIPS = N_SIMD * CLOCK_GHZ / 2.0  # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles at full rate
L2_PEAK_GBS = 34500.0  # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate
FRAME_BYTES = WIDTH * HEIGHT * 5   # BGR 8UC3 + depth 16UC1


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]


def step_stats(stamps, t0, window):
    """Per-step wall time from the completion times of the steps.  The steps are pipelined over `window` output slots whose
    results become ready in bursts (one per device lane), so single collect-to-collect intervals alternate between ~0 and a
    multiple of the step time; the statistics are taken over sliding windows of `window` consecutive steps, per step."""
    t = np.asarray([t0] + list(stamps))
    if len(t) <= window + 1:
        d = np.diff(t) * 1e3
    else:
        d = (t[window:] - t[:-window]) / window * 1e3
        d = d[len(d) // 10:]                      # the first tenth fills the pipeline
    return {"min": float(d.min()), "p10": float(pct(d, 0.1)), "median": float(np.median(d)), "p90": float(pct(d, 0.9)), "max": float(d.max()), "n": int(len(d)),
            "window_steps": int(window)}


def cpu_baseline(bank, frames, threshold, budget_s=12.0, min_frames=24):
    """The oracle (CPU restatement, kind 'port') single-threaded around the same boundary call the reference times
    (..._service.cpp:342-346): every frame timed on its own after 3 warm-ups, median / p10 / p90."""
    from oracle import oracle as o
    det = o.OracleDetector(bank)
    for w in range(3):
        det.match(frames[w % len(frames)], threshold)
    times = []
    t_start = time.perf_counter()
    while len(times) < min_frames or (time.perf_counter() - t_start < budget_s and len(times) < 16 * len(frames)):
        f = frames[len(times) % len(frames)]
        t0 = time.perf_counter()
        det.match(f, threshold)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port",
            "ms_per_frame": {"median": med * 1e3, "p10": pct(times, 0.1) * 1e3, "p90": pct(times, 0.9) * 1e3},
            "sample": "%d frames of the same batch timed one by one after 3 warm-ups, %d templates, single thread, %.1f s"
                      % (len(times), bank.num_templates(), time.perf_counter() - t_start)}


def cpu_baseline_all_cores(bank, frames, threshold, budget_s=8.0):
    """The same oracle on every host core the process may use (one detector per thread, frames dealt round robin; the ctypes
    calls release the GIL).  Informational: SURVEY 8(d) asks for the all-core figure next to the 1-core one."""
    import concurrent.futures as cf
    from oracle import oracle as o
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU share (cgroup v2 cpu.max = "<quota> <period>") is what it can really use
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    cores = min(cores, 64)
    dets = [o.OracleDetector(bank) for _ in range(cores)]
    t_end = time.perf_counter() + budget_s

    def work(i):
        n = 0
        while time.perf_counter() < t_end:
            dets[i].match(frames[(i + n * cores) % len(frames)], threshold)
            n += 1
        return n
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames over %d threads, %d templates, %.1f s" % (n, cores, bank.num_templates(), dt)}


def run_pipelined(det, k, B, threshold, uploads=None, stamps=None, collect_cap=1 << 16, collect=None):
    """k steps, software-pipelined over the context's output slots: the host finalisation (sort/unique) of a step overlaps the
    kernels of the following ones.  Exactly k enqueues and k collects; with `uploads` (a list of host batches) every step first
    uploads the next batch (fresh host frames: the transfer of step i+1 overlaps the kernels of step i)."""
    depth, inflight, out = det.max_outstanding, 0, None
    collect = collect or det.collect
    for i in range(k):
        if inflight == depth:
            out = collect(B, collect_cap)
            inflight -= 1
            if stamps is not None:
                stamps.append(time.perf_counter())
        if callable(uploads):
            uploads(det, i)
        elif uploads is not None:
            det.upload(uploads[i % len(uploads)])
        det.enqueue(B, threshold)
        inflight += 1
    while inflight:
        out = collect(B, collect_cap)
        inflight -= 1
        if stamps is not None:
            stamps.append(time.perf_counter())
    return out


def timed(torch, fn, sync_extra=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(t0)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0, t0


def secondary_line(torch, Detector, bank, frames, B, threshold, steps, overlap=True, uploads=None, async_input=False, max_candidates=0, collect_cap=1 << 16,
                   width=WIDTH, height=HEIGHT, breakdown=False, **det_kw):
    """One secondary workload on its own context: warm up, time `steps` pipelined steps, return {value, ms_per_step, ...}."""
    det = Detector(bank, width, height, device=torch.cuda.current_device(), max_batch=B, overlap=overlap, async_input=async_input,
                   max_candidates=max_candidates, **det_kw)
    if uploads is None:
        det.upload(frames)
    run_pipelined(det, 2 * det.max_outstanding + 2, B, threshold, uploads, None, collect_cap)
    stamps = []
    out, dt, t0 = timed(torch, lambda t0: run_pipelined(det, steps, B, threshold, uploads, stamps, collect_cap))
    st = det.stats()
    line = {"value": B * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "frames_per_step": B,
            "step_ms": step_stats(stamps, t0, det.max_outstanding), "matches_per_frame": float(np.mean([len(m) for m in out])),
            "coarse_candidates_per_frame": st["candidates"] / float(B)}
    if breakdown:   # per-kernel time with one step in flight (HIP events around every kernel)
        det.set_profiling(True)
        det.reset_profiling()
        for _ in range(3):
            det.enqueue(B, threshold)
            det.collect(B, collect_cap)
        line["kernel_ms_per_step"] = {k: v[0] / 3.0 for k, v in det.kernel_times().items() if v[1]}
        det.set_profiling(False)
    det.close()
    return line


def group_line(torch, bank, frames, B, threshold, steps, members, collective, host_batches=None):
    """The C++ device group (csrc/lmx_group.cpp) through dist.DeviceGroup: `members` members in ONE process.  On a one-GPU box the
    members share device 0 (peer-copy collective), so the device does `members` x the replicated pre-processing: the throughput is not a
    multi-GPU figure.  What transfers to a node with `members` GPUs is the HOST cost of driving them, reported as host_us_per_batch
    (time the calling thread spends inside upload + submit per batch; one host thread per member issues that member's launches)."""
    from linemod_pose_estimation_amd.dist import DeviceGroup
    g = DeviceGroup(bank, WIDTH, HEIGHT, members, devices=[torch.cuda.current_device()] * members, max_batch=B, collective=collective)
    g.upload(host_batches[0] if host_batches else frames)

    def run(k, stamps=None, host=None):
        inflight, out = 0, None
        for i in range(k):
            if inflight == g.depth:
                out = g.finish(B)
                inflight -= 1
                if stamps is not None:
                    stamps.append(time.perf_counter())
            t_a = time.perf_counter()
            if host_batches:
                g.upload(host_batches[i % len(host_batches)])
            g.submit(B, threshold)
            if host is not None:
                host.append(time.perf_counter() - t_a)
            inflight += 1
        while inflight:
            out = g.finish(B)
            inflight -= 1
            if stamps is not None:
                stamps.append(time.perf_counter())
        return out
    run(2 * g.depth + 2)
    stamps, host = [], []
    out, dt, t0 = timed(torch, lambda t0: run(steps, stamps, host))
    line = {"value": B * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "frames_per_step": B, "members": members,
            "collective": g.collective, "input": "fresh pageable host frames every batch, staged once and fanned out" if host_batches else "device-resident",
            "host_us_per_batch": {"median": float(np.median(host)) * 1e6, "p90": float(np.percentile(host, 90)) * 1e6},
            "step_ms": step_stats(stamps, t0, g.depth), "matches_per_frame": float(np.mean([len(m) for m in out]))}
    g.close()
    return line


def run_sharded(sm, k, B, threshold, uploads=None, stamps=None):
    """k steps over a ShardedMatcher, pipelined to its depth: the exchange and the host merge of a step overlap the kernels of the next ones.
    With `uploads` (a list of host batches) every step first uploads the next batch (each rank its frame group's share of it)."""
    inflight, out = 0, None
    for i in range(k):
        if inflight == sm.depth:
            out = sm.finish()
            inflight -= 1
            if stamps is not None:
                stamps.append(time.perf_counter())
        if uploads is not None:
            sm.upload(uploads[i % len(uploads)])
        sm.submit(B, threshold)
        inflight += 1
    while inflight:
        out = sm.finish()
        inflight -= 1
        if stamps is not None:
            stamps.append(time.perf_counter())
    return out


def timed_sharded(torch, dist, sm, steps, B, threshold, uploads=None):
    """Barrier + synchronize on both sides, MAX over ranks -- the contract's timing, for a secondary line of the multi-rank job."""
    run_sharded(sm, 2 * sm.depth + 2, B, threshold, uploads)
    dist.barrier()
    torch.cuda.synchronize()
    stamps = []
    t0 = time.perf_counter()
    out = run_sharded(sm, steps, B, threshold, uploads, stamps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    line = {"value": B * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "frames_per_step": B,
            "step_ms": step_stats(stamps, t0, sm.depth), "frame_groups": sm.G, "template_shards": sm.R, "gather_regrows": sm.regrows}
    if out is not None:
        line["matches_per_frame"] = float(np.mean([len(m) for m in out]))
    return line


def dist_extras(torch, dist, ShardedMatcher, Detector, synth, args, bank, frames, B, rank, world):
    """Secondary lines of the multi-rank job (every rank runs them, rank 0 reports): the host-frame boundary through the sharded path, and
    BASELINE configs[3] / [4] -- the 50 000-template bank over the N ranks of this job, strong scaling, in both decompositions."""
    extra = {}
    steps = max(20, min(args.steps, 60))
    try:   # the reference's boundary (fresh pageable host frames every step) at N ranks: every rank stages and transfers the batch over its own link
        perms = [np.random.default_rng(s_).permutation(B) for s_ in (1, 2, 3)]
        host_batches = [Detector.prepare_batch([[np.array(src, copy=True) for src in frames[i]] for i in p]) for p in perms]
        sm = ShardedMatcher(bank, WIDTH, HEIGHT, max_batch=B, overlap=not args.no_overlap)
        hf = timed_sharded(torch, dist, sm, steps, B, args.threshold, host_batches)
        hf["pcie_gbs_per_rank"] = hf["value"] * FRAME_BYTES / 1e9
        hf["input"] = "pageable host memory, %d bytes per frame, fresh frames every step on every rank (template shards: each rank transfers the whole batch)" % FRAME_BYTES
        extra["host_frames"] = hf
        del sm, host_batches
    except Exception as e:
        extra["host_frames"] = {"error": str(e)[:300]}
    try:
        bank50 = synth.make_bank(50000, seed=20250217)
        fr50 = [synth.make_scene(bank50, WIDTH, HEIGHT, seed=6000 + f, row_pad=0, texture=args.texture)[0] for f in range(B)]
        grids = sorted({1, world} | ({2, 4} & {g for g in (2, 4) if world % g == 0 and g < world}))
        strong = {}
        for G in grids:
            sm = ShardedMatcher(bank50, WIDTH, HEIGHT, max_batch=B, overlap=not args.no_overlap, frame_groups=G, hipgraph=(G == world and world > 1))
            sm.upload(fr50)
            ln = timed_sharded(torch, dist, sm, steps, B, args.threshold)
            st = sm.det.stats()
            ln["coarse_candidates_per_frame_and_rank"] = st["candidates"] / float(max(1, sm.frames_of(B)[1]))
            ln["per_rank"] = "%d frames x %d templates per step" % (sm.frames_of(B)[1], 50000 // sm.R)
            ln["hipgraph"] = bool(G == world and world > 1)
            strong["frame_groups_%d_x_template_shards_%d" % (G, world // G)] = ln
            del sm
        strong["workload"] = ("BASELINE configs[3]/[4]: the 50 000-template bank over the %d ranks of this job, %d device-resident 640x480 RGB-D frames per step, threshold %g; "
                              "value = whole-job frames/s (strong scaling: total work fixed). G x R = frame groups x template shards: 1 x N = every rank pre-processes all "
                              "frames and scores 50000/N templates (one frame's latency, configs[3]); N x 1 = every rank holds the whole bank and takes 1/N of the frames "
                              "(configs[4], hipGraph-captured chain)" % (world, B, args.threshold))
        extra["strong_50k"] = strong
    except Exception as e:
        extra["strong_50k"] = {"error": str(e)[:300]}
    return extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=64, help="frames per step (resident batch)")
    ap.add_argument("--templates", type=int, default=TEMPLATES_PER_GPU, help="templates per GPU")
    ap.add_argument("--threshold", type=float, default=THRESHOLD)
    ap.add_argument("--texture", type=float, default=0.6, help="background texture amplitude of the synthetic scenes (synth.make_scene)")
    ap.add_argument("--hipgraph", action="store_true", help="replay the per-batch kernel chain as one hipGraph (LMX_CTX_HIPGRAPH)")
    ap.add_argument("--no-overlap", action="store_true", help="one device lane (LMX_CTX_OVERLAP off) and two steps in flight")
    ap.add_argument("--torch-stream", action="store_true", help="run on torch's current stream instead of a private one")
    ap.add_argument("--sharded", action="store_true", help="use the N>1 code path (ShardedMatcher + all-gather) even with one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workload lines (host frames, busy scene, low threshold)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config lines (config 0 / 3 / 4 / 5 shapes, device group) inside `extra`")
    ap.add_argument("--no-events", action="store_true", help="no HIP events in the timed region (roofline then uses the untimed pass)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on stdout when the process group is
    # created) get stderr as their fd 1 for the rest of the run, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); it has to be in the environment
    # before the runtime starts.  (Earlier in round 3 the sharded path measured 126 k frames/s at 4 queues and 130-132 k at 8-12: that was
    # RCCL's lazy set-up order, fixed since -- ShardedMatcher runs the communicator's first collective before it creates its context.)
    # Late in round 3 ("stream placement"): 8 queues give the best rate when the detector's streams are the first in the
    # process, but with 2 or 3 other used streams alive the lanes land badly (-7 %); with the runtime's default of 4 there is no bad placement.
    # One rank: a dedicated process, 8.  Several ranks: torch.distributed and RCCL bring streams of their own, whose number this script does
    # not control: 4 (one rank, RCCL in the loop: 140.6 k frames/s at 4 against 141.6 k at 8 in the good placement).
    # Under `rocprofv3 ... -- python3 bench.py` the profiler's preloaded library has started the HIP runtime before this line runs, so the
    # variable only counts when it is already in the environment of the command (scripts/profile_r02.sh exports it); the line records which.
    hwq_preset = "GPU_MAX_HW_QUEUES" in os.environ
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8" if int(os.environ.get("WORLD_SIZE", "1")) <= 1 else "4")
    hwq = {"GPU_MAX_HW_QUEUES": int(os.environ["GPU_MAX_HW_QUEUES"]), "set_by": "environment of the command" if hwq_preset else "bench.py before the runtime starts",
           "profiler_preloaded": any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or "rocprof" in os.environ.get("LD_PRELOAD", "")}
    if hwq["profiler_preloaded"] and not hwq_preset:
        hwq["note"] = "a profiler started the runtime before bench.py could set the variable: the runtime default (4 queues) is in effect"
    import torch
    import torch.distributed as dist
    from linemod_pose_estimation_amd import synth, Detector, PinnedArena
    from linemod_pose_estimation_amd.dist import ShardedMatcher

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch N>1 with torch.distributed.run" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the matching path has no CPU implementation outside the test oracle")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.sharded
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_total = args.templates * world
    bank = synth.make_bank(n_total, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
    frames = [synth.make_scene(bank, WIDTH, HEIGHT, seed=3000 + f, row_pad=0, texture=args.texture)[0] for f in range(args.frames)]
    B = args.frames

    if not use_dist:
        det = Detector(bank, WIDTH, HEIGHT, device=local_rank, max_batch=B,
                       stream=(torch.cuda.current_stream().cuda_stream if args.torch_stream else None), hipgraph=args.hipgraph,
                       overlap=not args.no_overlap)
        det.upload(frames)

        def step():
            det.enqueue(B, args.threshold)
            return det.collect(B)

        def run_steps(k, stamps=None):
            return run_pipelined(det, k, B, args.threshold, None, stamps)
        raw_det = det
    else:
        sm = ShardedMatcher(bank, WIDTH, HEIGHT, max_batch=B, overlap=not args.no_overlap)
        sm.upload(frames)

        def step():
            return sm.step(B, args.threshold)

        def run_steps(k, stamps=None):
            """the same software pipeline over the sharded path: exchange + host merge of a step overlap the next steps' kernels"""
            return run_sharded(sm, k, B, args.threshold, None, stamps)
        raw_det = sm.det

    # 1. untimed pass, one step in flight, HIP events around every kernel: per-kernel breakdown and the dominant kernel's name
    for _ in range(2):
        out = step()
    raw_det.set_profiling(True)
    raw_det.reset_profiling()
    n_prof = 4
    for _ in range(n_prof):
        out = step()
    breakdown = {k: v[0] / float(n_prof) for k, v in raw_det.kernel_times().items()}  # ms per step
    raw_det_launches = {k: v[1] // n_prof for k, v in raw_det.kernel_times().items()}
    dom = max(breakdown, key=breakdown.get)
    # from here on exactly the timed configuration: events only around the dominant kernel (each timed launch adds two event
    # records to its stream)
    raw_det.set_profiling(False if args.no_events else dom)
    # 2. untimed priming, pipelined like the timed steps.  The first concurrent uses of the device lanes / the communication
    # stream block the submitting thread for 5-7 ms a few times per process while the HIP runtime grows its signal pools
    # (seen inside hipMemcpyAsync/hipMemsetAsync of the matching stage; gone after ~20 steps, scripts/sharded_experiment.py);
    # that one-off cost must not land in the timed steps whatever W is.  3. the W warm-up steps.
    out = run_steps(48)
    if args.warmup > 0:
        out = run_steps(args.warmup)
    raw_det.reset_profiling()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    stamps = []
    t0 = time.perf_counter()
    out = run_steps(args.steps, stamps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    raw_det.set_profiling(False)
    dist_extra = None
    line = None

    if rank == 0:
        times = raw_det.kernel_times()
        dom_ms, dom_n = times[dom]
        alg = raw_det.algorithmic_bytes(dom, B)
        if not dom_n:  # --no-events: fall back to the untimed profiling pass
            launches_per_step = max(1, raw_det_launches.get(dom, 1))
            dom_ms, dom_n = breakdown[dom] * args.steps, launches_per_step * args.steps
        lps = dom_n / float(args.steps)             # launches of the dominant kernel per step
        excl_ms = breakdown[dom] / max(1, raw_det_launches.get(dom, 1))   # one step in flight: the kernel by itself
        dev_name = raw_det.device_kernel_name(dom)
        # PMC evidence of the SAME workload (scripts/profile_r02.sh -> scripts/pmc_summary2.py, committed under profiles/), keyed by
        # device kernel and grid size; values are per launch.  Only used when the workload matches.
        pmc, pmc_src = None, None
        ppath = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(ppath):
            pj = json.load(open(ppath))
            wl = pj.get("workload", {})
            if (wl.get("frames") == B and wl.get("templates") == args.templates and wl.get("threshold") == args.threshold and wl.get("texture") == args.texture and
                    bool(wl.get("score_no_prune", False)) == (os.environ.get("LMX_SCORE_NO_PRUNE", "0") not in ("", "0"))):
                cands = [r for r in pj["kernels"].values() if r["kernel"] == dev_name]
                if cands:
                    pmc = max(cands, key=lambda r: r.get("duration_us_one_lane_trace") or 0.0)   # the level / launch shape that dominates
                    pmc_src = "profiles/pmc_summary.json (%s)" % pj.get("source", "?")
        hbm_achieved = (alg / lps) / (excl_ms * 1e-3) / 1e9   # algorithmic bytes per launch / exclusive launch time
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        valu = pmc["counters"].get("SQ_INSTS_VALU") if pmc else None
        salu = pmc["counters"].get("SQ_INSTS_SALU") if pmc else None
        l2req = pmc["counters"].get("TCP_TCC_READ_REQ_sum") if pmc else None
        valu_view = None
        if valu:
            gips = valu / (excl_ms * 1e-3) / 1e9   # G wave64-VALU instructions per second
            valu_view = {"achieved": gips, "peak": VALU_PEAK_GIPS, "unit": "G wave64 VALU instr/s", "frac": gips / VALU_PEAK_GIPS,
                         "peak_definition": "%d SIMD-32 x %.1f GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md, 'Wave scheduling'); v_perm / v_alignbit / DPP / "
                                            "v_add3 / 24-bit multiplies issue at ~4.2 cycles, plain adds / ands / shifts at ~2.4 (profiles/r02_valu_issue_microbench.txt)" % (N_SIMD, CLOCK_GHZ),
                         "valu_instructions_per_launch": valu, "salu_instructions_per_launch": salu,
                         "salu_issue_frac": (salu / (excl_ms * 1e-3) / 1e9) / (256 * CLOCK_GHZ) if salu else None}
        if l2req:
            # the scoring kernel gathers from memories that are resident in the XCDs' L2s: what it runs into is the L2 -> L1 fill
            # bandwidth (128-byte lines; 1 L2 miss = 128 B of FETCH_SIZE x 2 in the same profile)
            achieved = l2req * 128.0 / (excl_ms * 1e-3) / 1e9
            roofline = {"bound": "l2", "kernel": dev_name, "achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBS, "traffic": traffic,
                        "peak_definition": "aggregate L2 bandwidth of the 8 XCDs, MI355X_MICROARCH.md 'L2 (per XCD)': ~34.5 TB/s; the same guide measures 16.8-18.8 TB/s "
                                           "for L2-resident row gathers, the access pattern of this kernel",
                        "l2_read_requests_per_launch": l2req, "line_bytes": 128, "counters_from": pmc_src}
            vmem = pmc["counters"].get("SQ_INSTS_VMEM_RD")
            if vmem:
                # end of round 2 (profiles/r02s_score_load_experiments.txt): with half the line fills the kernel is 3 % faster, with half the
                # loads 23 %: its time follows the bytes the vector L1s hand to the lanes (one 64-lane dword load = 256 B), not the fills
                l1 = vmem * 256.0 / (excl_ms * 1e-3) / 1e9
                l1_peak = 256 * 64 * CLOCK_GHZ
                roofline["l1_delivery"] = {"achieved": l1, "peak": l1_peak, "unit": "GB/s", "frac": l1 / l1_peak, "wave_loads_per_launch": vmem,
                                           "peak_definition": "256 CUs x 64 B/clk x %.1f GHz (vector L1 -> VGPR return path, nominal); the loads are dword-aligned, "
                                                              "not line-aligned: a wave's 256 B straddle three lines" % CLOCK_GHZ,
                                           "evidence": "profiles/r02s_score_load_experiments.txt"}
        elif valu_view:
            roofline = dict(valu_view)
            roofline.update({"bound": "valu_issue", "kernel": dev_name, "traffic": traffic, "counters_from": pmc_src})
        else:
            roofline = {"bound": "hbm", "kernel": dev_name, "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "note": "no PMC summary for this workload under profiles/: algorithmic-bytes view only (cache-resident working set, "
                                                    "not a utilisation)"}
        if valu_view and roofline.get("bound") != "valu_issue":
            roofline["valu_issue"] = valu_view
        if pmc is not None:
            # the other kernels of the step from the same committed PMC run (profiles/pmc_summary.json; per launch, one lane in flight): what each
            # is bound by in one place -- the quantisers by VALU issue, the spread by its scattered stores, none of them by HBM (peak 8000 GB/s)
            roofline["other_kernels"] = {
                "%s@%d" % (r["kernel"], r["grid_size"]): {"duration_us": round(r.get("duration_us_one_lane_trace") or 0.0, 1), "valu_issue_frac_2cyc": round(r.get("valu_issue_frac_2cyc") or 0.0, 3),
                                                            "hbm_gbs": round(r.get("hbm_gbs") or 0.0), "l2_to_l1_tbs": round(r.get("l2_to_l1_tbs_128B") or 0.0, 2),
                                                            "waves_per_simd": round(r.get("avg_waves_per_simd") or 0.0, 1),
                                                            "valu_per_wave": round((r.get("per_wave") or {}).get("valu") or 0.0),
                                                            # the quantisers' waves each produce 8 output pixels per lane (64 x 32 tile, 256 threads)
                                                            **({"valu_per_output_pixel": round(((r.get("per_wave") or {}).get("valu") or 0.0) / 8.0, 1)} if r["kernel"] in ("k_color_quantize", "k_depth_quantize") else {})}
                for r in pj["kernels"].values() if r["kernel"].startswith("k_") and r["kernel"] != dev_name and (r.get("duration_us_one_lane_trace") or 0.0) >= 3.0}
        roofline.update({
            "avg_launch_ms_exclusive": excl_ms, "avg_launch_ms_timed_region": dom_ms / dom_n, "launches_per_step": lps,
            # HBM view of the same kernel: algorithmic bytes per launch, what they would need of HBM, what HBM really moved
            "hbm": {"algorithmic_bytes_per_launch": alg / lps, "algorithmic_gbs": hbm_achieved, "algorithmic_frac_of_peak": hbm_achieved / HBM_PEAK_GBS,
                    "measured_bytes_per_launch": traffic, "measured_gbs": (traffic / (excl_ms * 1e-3) / 1e9) if traffic else None,
                    "measured_frac_of_peak": (traffic / (excl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None, "peak": HBM_PEAK_GBS}})
        fps = B * args.steps / dt
        value = fps * (n_total / float(TEMPLATES_PER_GPU))
        dens = {"cg_l0": float((raw_det.debug_quantized(0, 0, 0) != 0).mean()), "cg_l1": float((raw_det.debug_quantized(0, 1, 0) != 0).mean()),
                "dn_l0": float((raw_det.debug_quantized(0, 0, 1) != 0).mean())}
        st = raw_det.stats()
        line = {
            "metric": "rgbd_frames_per_sec_matched",
            "value": value,
            "unit": "frames/s (640x480 RGB-D, per 3000-template bank)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "step_ms": step_stats(stamps, t0, raw_det.max_outstanding),
            "frames_per_sec": fps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 640x480 RGB-D, ColorGradient+DepthNormal, T={5,8}, %d templates/GPU" % args.templates,
                       "input": "device-resident (frames uploaded once before the timed region; host-frame rates: host_frames*)",
                       "frames_per_step": B, "templates_per_gpu": args.templates, "templates_total": n_total,
                       "threshold": args.threshold, "device_lanes": raw_det.max_outstanding // 2, "hipgraph": bool(args.hipgraph), "hw_queues": hwq,
                       "parallelism": "template-shard x%d + all-gather" % world,
                       "matches_per_frame": float(np.mean([len(m) for m in out])),
                       "coarse_candidates_per_frame": st["candidates"] / float(B), "scene_texture": args.texture,
                       "score_no_prune": os.environ.get("LMX_SCORE_NO_PRUNE", "0") not in ("", "0"),
                       "label_density": dens},
            "roofline": roofline,
            "kernel_ms_per_step": breakdown,
            "template_cells_per_sec": value * TEMPLATES_PER_GPU * 1200.0,   # N x 1200 coarse cells x frames/s
        }
        if not use_dist:
            det.close()
    if use_dist and not args.no_extra:
        # The secondary lines of the multi-rank job run on EVERY rank (collectives inside) and have never met more than one rank on hardware: the
        # headline above must not depend on them.  It is complete at this point (rank 0 holds it); if the secondary lines have not finished after
        # LMX_BENCH_EXTRA_DEADLINE_S seconds (a rank that failed alone would leave the others inside a collective), rank 0 prints the line without
        # them and every rank leaves.
        import threading
        comm = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(), "ranks_per_node": int(os.environ.get("LOCAL_WORLD_SIZE", str(world))),
                "note": "what the process group reports; every collective of this line ran over it (torch backend nccl = RCCL on ROCm)"}
        deadline_s = float(os.environ.get("LMX_BENCH_EXTRA_DEADLINE_S", "600"))
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0 and line is not None:
                line["rccl_world"] = comm["rccl_world"]
                line["extra"] = {"error": "the secondary lines of the multi-rank job did not finish within %.0f s; headline only" % deadline_s, "communicator": comm}
                os.write(json_fd, (json.dumps(line) + "\n").encode())
            os._exit(0)
        watchdog = threading.Timer(deadline_s, bail)
        watchdog.daemon = True
        watchdog.start()
        dist_extra = dist_extras(torch, dist, ShardedMatcher, Detector, synth, args, bank, frames, B, rank, world)
        dist_extra["communicator"] = comm
        finished.set()
        watchdog.cancel()
    if rank == 0:
        if not args.no_extra and world == 1 and not use_dist:
            # the boundary the reference actually has: fresh host frames on every call.  Three distinct batches (the 64 scenes in
            # three orders, separate host arrays) rotate, so every step transfers data that is not on the device yet.
            perms = [np.random.default_rng(s).permutation(B) for s in (1, 2, 3)]
            host_batches = [[[np.array(src, copy=True) for src in frames[i]] for i in p] for p in perms]
            hsteps = max(40, min(args.steps, 100))
            # descriptors (pointer, size, stride per image) built once per batch, as a C++ caller holding cv::Mat headers would have them
            hf = secondary_line(torch, Detector, bank, None, B, args.threshold, hsteps, overlap=not args.no_overlap, uploads=[Detector.prepare_batch(b) for b in host_batches])
            hf["pcie_gbs"] = hf["value"] * FRAME_BYTES / 1e9
            hf["input"] = "pageable host memory (numpy), %d bytes per frame, fresh frames every step" % FRAME_BYTES
            line["host_frames"] = hf
            arena = PinnedArena(3 * B * (FRAME_BYTES + 1024))
            pinned_batches = [[[arena.put(src) for src in fr] for fr in batch] for batch in host_batches]
            hp = secondary_line(torch, Detector, bank, None, B, args.threshold, hsteps, overlap=not args.no_overlap, uploads=[Detector.prepare_batch(b) for b in pinned_batches],
                                async_input=True)
            hp["pcie_gbs"] = hp["value"] * FRAME_BYTES / 1e9
            hp["input"] = "pinned host memory (lmx_host_alloc), DMA straight from the caller's buffers (LMX_CTX_ASYNC_INPUT)"
            line["host_frames_pinned"] = hp
            del pinned_batches
            arena.close()
            extra = {}
            # the same boundary one step earlier: what the reference's node holds before match() is the Ensenso's MONO8 752x480
            # image and a sensor-size depth image; detect_cb makes the BGR 640x480 frame on the host (..._service.cpp:293-326).  lmx_ctx_upload_raw
            # takes the raw pair: 1.08 MB over PCIe per frame instead of 1.54, MONO->BGR + 3x3 blur + crop on the device
            try:
                SW_, CX_ = 752, 56
                raw_batches = []
                for pm in perms:
                    rb = []
                    for i in pm:
                        mono_ = np.ascontiguousarray(np.pad(frames[i][0][:, :, 1], ((0, 0), (CX_, SW_ - WIDTH - CX_)), mode="edge"))
                        depth_ = np.ascontiguousarray(np.pad(frames[i][1], ((0, 0), (CX_, SW_ - WIDTH - CX_)), mode="edge"))
                        rb.append([mono_, depth_])
                    raw_batches.append(Detector.prepare_batch(rb))

                def up_raw_rgbd(det, i):
                    det.upload_raw(raw_batches[i % len(raw_batches)], (SW_, HEIGHT), (CX_, 0), blur3=True, mono=True)
                hr = secondary_line(torch, Detector, bank, None, B, args.threshold, hsteps, overlap=not args.no_overlap, uploads=up_raw_rgbd)
                hr["pcie_gbs"] = hr["value"] * SW_ * HEIGHT * 3 / 1e9
                hr["input"] = "raw camera frames in pageable host memory, fresh every step: MONO8 752x480 + u16 depth 752x480 (%d bytes per frame); MONO->BGR, 3x3 blur, crop on the device" % (SW_ * HEIGHT * 3)
                line["host_frames_raw_camera"] = hr
                del raw_batches
            except Exception as e:
                line["host_frames_raw_camera"] = {"error": str(e)[:300]}
            busy = [synth.make_scene(bank, WIDTH, HEIGHT, seed=3000 + f, row_pad=0, texture=1.0)[0] for f in range(B)]
            extra["busy_scene"] = secondary_line(torch, Detector, bank, busy, B, args.threshold, max(40, min(args.steps, 100)), overlap=not args.no_overlap)
            extra["busy_scene"]["scene_texture"] = 1.0
            # The data-INDEPENDENT leg of the scoring kernel.  The headline's k_score_coarse prunes exactly: on these scenes the
            # average wave stops after half its loads, so its time depends on the data.  With LMX_SCORE_NO_PRUNE=1 (read when the context is created)
            # the early exits are compiled out and the kernel does similarity()'s full work -- every feature at every placement, the
            # same candidates -- which bounds the kernel from below whatever the scene and the threshold.
            try:
                os.environ["LMX_SCORE_NO_PRUNE"] = "1"
                try:
                    fw = secondary_line(torch, Detector, bank, frames, B, args.threshold, max(40, min(args.steps, 100)), overlap=not args.no_overlap, breakdown=True)
                finally:
                    del os.environ["LMX_SCORE_NO_PRUNE"]
                k_us = fw["kernel_ms_per_step"]["k_score_coarse"] * 1e3
                fw["score_kernel_us"] = k_us
                alg_gbs = alg / lps / (k_us * 1e-6) / 1e9
                fw["algorithmic"] = {"bytes_per_launch": alg / lps, "gbs": alg_gbs, "frac_of_l2_peak": alg_gbs / L2_PEAK_GBS, "frac_of_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                                     "note": "SURVEY 8(d)'s B_score (1 byte per feature and placement + the score maps) / the kernel's time with one step in flight; the kernel "
                                             "reads the memories nibble-packed (half a byte per response) out of the XCDs' L2s, so the L2 peak is the bound that applies"}
                fwp = os.path.join(ROOT, "profiles", "pmc_summary_full_work.json")
                if os.path.exists(fwp):
                    pj2 = json.load(open(fwp))
                    wl2 = pj2.get("workload", {})
                    rows2 = [r for r in pj2["kernels"].values() if r["kernel"] == dev_name]
                    if rows2 and wl2.get("score_no_prune") and wl2.get("frames") == B and wl2.get("templates") == args.templates and wl2.get("threshold") == args.threshold and wl2.get("texture") == args.texture:
                        r2 = max(rows2, key=lambda r: r.get("duration_us_one_lane_trace") or 0.0)
                        c2 = r2["counters"]
                        if c2.get("TCP_TCC_READ_REQ_sum"):
                            tb = c2["TCP_TCC_READ_REQ_sum"] * 128.0 / (k_us * 1e-6) / 1e9
                            fw["l2"] = {"achieved": tb, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": tb / L2_PEAK_GBS, "l2_read_requests_per_launch": c2["TCP_TCC_READ_REQ_sum"]}
                        if c2.get("SQ_INSTS_VMEM_RD"):
                            l1 = c2["SQ_INSTS_VMEM_RD"] * 256.0 / (k_us * 1e-6) / 1e9
                            fw["l1_delivery"] = {"achieved": l1, "peak": 256 * 64 * CLOCK_GHZ, "unit": "GB/s", "frac": l1 / (256 * 64 * CLOCK_GHZ), "wave_loads_per_launch": c2["SQ_INSTS_VMEM_RD"],
                                                 "wave_loads_per_wave": c2["SQ_INSTS_VMEM_RD"] / c2["SQ_WAVES"] if c2.get("SQ_WAVES") else None}
                        if c2.get("SQ_INSTS_VALU"):
                            fw["valu_issue_frac_2cyc"] = c2["SQ_INSTS_VALU"] / (k_us * 1e-6) / 1e9 / VALU_PEAK_GIPS
                        fw["hbm_measured_bytes_per_launch"] = r2.get("hbm_bytes_per_launch")
                        fw["rocprofv3_one_lane_kernel_us"] = r2.get("duration_us_one_lane_trace")
                        fw["counters_from"] = "profiles/pmc_summary_full_work.json (%s)" % pj2.get("source", "?")
                fw["workload"] = "the headline workload with LMX_SCORE_NO_PRUNE=1: k_score_coarse reads every feature at every placement (no early exit); identical candidates and matches"
                extra["score_full_work"] = fw
            except Exception as e:
                extra["score_full_work"] = {"error": str(e)[:300]}
            lowB = 2
            try:
                extra["low_threshold"] = secondary_line(torch, Detector, bank, frames[:lowB], lowB, 50.0, 6, overlap=False, max_candidates=1 << 21, collect_cap=1 << 22)
                extra["low_threshold"]["threshold"] = 50.0
            except Exception as e:  # e.g. candidate capacity exceeded: report, do not lose the line
                extra["low_threshold"] = {"error": str(e)[:200], "threshold": 50.0}
            if not args.no_configs:
                # the 1-GPU part of every other BASELINE.json config, each on its own context, timed like the headline (pipelined steps,
                # device lanes); scenes and banks follow SURVEY 8(d)'s generator with seed = 20250213 + config index
                csteps = max(20, min(args.steps, 60))
                try:   # configs[0]: ColorGradient only, ~3000 templates, through the host-frame boundary (fresh pageable frames every step)
                    bank0 = synth.make_bank(args.templates, modalities=("ColorGradient",), T=(5, 8), seed=20250214)
                    fr0 = [synth.make_scene(bank0, WIDTH, HEIGHT, seed=4000 + f, row_pad=0, texture=args.texture)[0] for f in range(B)]
                    hb0 = [[[np.array(src, copy=True) for src in fr0[i]] for i in p] for p in perms]
                    c0 = secondary_line(torch, Detector, bank0, None, B, args.threshold, csteps, uploads=[Detector.prepare_batch(b) for b in hb0])
                    c0["workload"] = "BASELINE configs[0] shape on the GPU: 640x480, ColorGradient only, %d templates, fresh pageable host frames every step" % args.templates
                    c0["pcie_gbs"] = c0["value"] * WIDTH * HEIGHT * 3 / 1e9
                    c0["resident"] = secondary_line(torch, Detector, bank0, fr0, B, args.threshold, csteps)
                    # the reference's real camera for this config is the Ensenso: MONO8 752x480, which detect_cb turns into BGR, blurs 3x3 and crops
                    # to 640x480 on the host (..._service.cpp:293-326).  lmx_ctx_upload_raw takes the raw frame and does those steps on the
                    # device: 0.36 MB over PCIe per frame instead of 0.92
                    mono = [np.ascontiguousarray(np.pad(f[0][:, :, 1], ((0, 0), (56, 56)), mode="edge")) for f in fr0]
                    mono_batches = [Detector.prepare_batch([[np.array(mono[i], copy=True)] for i in p]) for p in perms]

                    def up_raw(det, i):
                        det.upload_raw(mono_batches[i % len(mono_batches)], (752, 480), (56, 0), blur3=True, mono=True)
                    craw = secondary_line(torch, Detector, bank0, None, B, args.threshold, csteps, uploads=up_raw)
                    craw["pcie_gbs"] = craw["value"] * 752 * 480 / 1e9
                    craw["workload"] = "the same bank, raw MONO8 752x480 camera frames (fresh pageable frames every step): MONO->BGR, 3x3 blur and crop on the device"
                    c0["raw_mono_752x480"] = craw
                    # configs[0] as the reference runs it: the service node reads <object>_templates.yml and builds its detector on EVERY request
                    # (..._service.cpp:1784-1786), then matches one frame.  Through the C ABI's caches: the yml parse, the binary
                    # side file, the in-process bank cache, the device-context cache, and the whole request when all of them are warm
                    try:
                        import ctypes as C
                        import tempfile
                        from linemod_pose_estimation_amd import _lib, NativeBank
                        from linemod_pose_estimation_amd.detector import _images, MATCH_DTYPE
                        L = _lib.lib()
                        tdir = tempfile.mkdtemp(prefix="lmx_bench_yml_")
                        yml = os.path.join(tdir, "memoryChip2_ensenso_templates.yml")
                        NativeBank.from_bank(bank0).save_yaml(yml)

                        def ms(fn):
                            t_ = time.perf_counter()
                            r_ = fn()
                            return (time.perf_counter() - t_) * 1e3, r_
                        hb = C.c_void_p()
                        t_parse, _ = ms(lambda: _lib.check(L.lmx_bank_load_yaml(yml.encode(), C.byref(hb))))
                        L.lmx_bank_destroy(hb)
                        h1 = C.c_void_p()
                        t_first, _ = ms(lambda: _lib.check(L.lmx_bank_load_yaml_cached(yml.encode(), C.byref(h1))))       # parse + write <yml>.lmxcache
                        desc = _lib.CtxDesc(local_rank, WIDTH, HEIGHT, 1, 0, 0, 1, None, 0)
                        ctx, hit = C.c_void_p(), C.c_int32()
                        t_ctx_cold, _ = ms(lambda: _lib.check(L.lmx_ctx_acquire(h1, C.byref(desc), C.byref(ctx), C.byref(hit))))
                        src = [np.array(a, copy=True) for a in fr0[0]]
                        imgs, keep = _images([src])
                        outm = np.zeros(1 << 14, MATCH_DTYPE)
                        nm = C.c_size_t()
                        _lib.check(L.lmx_match(ctx, imgs, 1, C.c_float(args.threshold), None, 0, outm.ctypes.data, len(outm), C.byref(nm)))
                        L.lmx_ctx_unref(ctx)

                        def request():
                            hq, cq, hitq = C.c_void_p(), C.c_void_p(), C.c_int32()
                            _lib.check(L.lmx_bank_load_yaml_cached(yml.encode(), C.byref(hq)))
                            _lib.check(L.lmx_ctx_acquire(hq, C.byref(desc), C.byref(cq), C.byref(hitq)))
                            _lib.check(L.lmx_match(cq, imgs, 1, C.c_float(args.threshold), None, 0, outm.ctypes.data, len(outm), C.byref(nm)))
                            L.lmx_ctx_unref(cq)
                            L.lmx_bank_release(hq)
                            return hitq.value
                        for _ in range(20):
                            request()
                        ts = []
                        for _ in range(200):
                            t_req, was_hit = ms(request)
                            ts.append(t_req * 1e3)
                        c0["yml_request_flow"] = {
                            "yml_mb": os.path.getsize(yml) / 1e6, "lmxcache_mb": os.path.getsize(yml + ".lmxcache") / 1e6, "yml_parse_ms": t_parse,
                            "first_cached_load_ms": t_first, "context_build_ms": t_ctx_cold, "context_cache_hit": int(was_hit),
                            "warm_request_us": {"median": float(np.median(ts)), "p10": pct(ts, 0.1), "p90": pct(ts, 0.9), "n": len(ts)},
                            "note": "per request: lmx_bank_load_yaml_cached + lmx_ctx_acquire + lmx_match (one fresh host frame) + unref + release, all caches warm; "
                                    "the reference re-parses the yml and rebuilds the detector every time"}
                        L.lmx_bank_release(h1)
                        L.lmx_cache_trim()   # no idle context may stay behind: it would cost the lines below 4-8 % (scripts/idle_context_effect.py)
                        del keep
                    except Exception as e:
                        c0["yml_request_flow"] = {"error": str(e)[:300]}
                    extra["config0_cg_only"] = c0
                    del bank0, fr0, hb0, mono, mono_batches
                except Exception as e:
                    extra["config0_cg_only"] = {"error": str(e)[:300]}
                try:   # configs[2]: 1280x1024 -> the 1280x960 crop (T=5 does not divide 1024: upstream would assert), 2 classes x 3000 templates
                    bank3 = synth.make_bank(3000, seed=20250216, classes=["cpu_binary", "memoryChip2"])
                    B3 = 16
                    fr3 = [synth.make_scene(bank3, 1280, 960, seed=5000 + f, row_pad=0, texture=args.texture, n_instances=8, n_distractors=12)[0] for f in range(B3)]
                    c3 = secondary_line(torch, Detector, bank3, fr3, B3, args.threshold, csteps, width=1280, height=960)
                    c3["workload"] = "BASELINE configs[2]: 1280x960 crop of 1280x1024 RGB-D, 2 classes x 3000 templates, device-resident frames"
                    extra["config3_1280x960_2x3000"] = c3
                    del bank3, fr3
                except Exception as e:
                    extra["config3_1280x960_2x3000"] = {"error": str(e)[:300]}
                try:   # configs[3] / [4]: the 50k-template bank over 8 GPUs, the part ONE GPU runs = rank 3's 6250-template shard
                    bank50 = synth.make_bank(50000, seed=20250217)
                    fr50 = [synth.make_scene(bank50, WIDTH, HEIGHT, seed=6000 + f, row_pad=0, texture=args.texture)[0] for f in range(B)]
                    c4 = secondary_line(torch, Detector, bank50, fr50, B, args.threshold, csteps, shard_rank=3, shard_world=8)
                    c4["workload"] = "BASELINE configs[3], per-GPU part: rank 3 of 8 of the 50 000-template bank (6250 templates), %d resident frames per step, eager lanes" % B
                    extra["config4_shard_6250"] = c4
                    c5 = secondary_line(torch, Detector, bank50, fr50, B, args.threshold, csteps, shard_rank=3, shard_world=8, hipgraph=True)
                    c5["workload"] = "BASELINE configs[4], per-GPU part: 64 concurrent frames x the 6250-template shard, hipGraph-captured chain replayed on three device lanes"
                    extra["config5_shard_6250_hipgraph_lanes"] = c5
                    # The same (frame, template) pairs per GPU in the other decompositions of the G x R grid (frame groups x template shards, 8 GPUs): what ONE
                    # GPU runs per 64-frame step of the job.  8 x 1 = frame sharding: the whole 50 000-template bank resident, 8 of the 64 frames, no replicated
                    # pre-processing.  `job_frames_per_sec_8gpu` = 64 frames / this GPU's step time (the exchange is one small all-gather off the critical path).
                    proxy = {"1x8": {"ms_per_step": c5["ms_per_step"], "frames": B, "templates": 6250, "job_frames_per_sec_8gpu": B / c5["ms_per_step"] * 1e3}}
                    for G_ in (2, 4, 8):
                        R_ = 8 // G_
                        nf_ = B // G_
                        ln_ = secondary_line(torch, Detector, bank50, fr50[:nf_], nf_, args.threshold, csteps, shard_rank=R_ // 2, shard_world=R_, hipgraph=True, breakdown=(G_ == 8))
                        ln_["workload"] = "per-GPU part of %d frame groups x %d template shards: %d frames x %d templates per step, hipGraph lanes" % (G_, R_, nf_, 50000 // R_)
                        proxy["%dx%d" % (G_, R_)] = {"ms_per_step": ln_["ms_per_step"], "frames": nf_, "templates": 50000 // R_, "job_frames_per_sec_8gpu": B / ln_["ms_per_step"] * 1e3}
                        if G_ == 8:
                            extra["config5_frames8_x_50k"] = ln_
                    extra["strong_50k_per_gpu_proxy"] = {"grids_frame_groups_x_template_shards": proxy,
                                                         "note": "one GPU's share of a 64-frame step over the 50 000-template bank on 8 GPUs, per decomposition; measured on ONE GPU, "
                                                                 "not a multi-GPU figure (the N > 1 job reports extra.strong_50k)"}
                    del bank50, fr50
                except Exception as e:
                    extra["config4_shard_6250"] = {"error": str(e)[:300]}
                try:   # the realistic bank: neighbouring views of ONE object rendered from the reference's own mesh (meshsynth.py), scenes with rendered chips
                    from linemod_pose_estimation_amd import meshsynth as ms
                    mbank, _, _, _ = ms.load_bank("memoryChip2")
                    chip, cpu_mesh, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
                    distinct = [ms.make_scene(chip, views, seed=7000 + f, n_instances=3, other_tri=cpu_mesh, n_other=2, texture=args.texture)[0] for f in range(16)]
                    mframes = [distinct[f % len(distinct)] for f in range(B)]
                    mb = secondary_line(torch, Detector, mbank, mframes, B, args.threshold, csteps, breakdown=True)
                    mb["workload"] = ("2652 templates trained from rendered views of the reference's memoryChip2.stl over its own view grid (26 directions x 6 distances x 17 "
                                      "in-plane rotations), %d frames per step (16 distinct scenes: 3 rendered chips + 2 cpu_binary distractors on texture), threshold %g" % (B, args.threshold))
                    mb["threshold_85"] = secondary_line(torch, Detector, mbank, mframes, B, 85.0, max(10, csteps // 2), max_candidates=1 << 16)
                    # the consumer of `matches` on the device: lmx_ctx_collect_clusters instead of lmx_ctx_collect, with the bank's
                    # renderer-params side-car (rects, distances); next to it the host chain (collect + lmx_cluster_matches per frame)
                    from linemod_pose_estimation_amd.detector import cluster_matches
                    _, mrects, mdists, _ = ms.load_bank("memoryChip2")
                    side = (8, ms.ENSENSO["radius_min"], ms.ENSENSO["radius_step"], 2)
                    dcl = Detector(mbank, WIDTH, HEIGHT, device=local_rank, max_batch=B, overlap=True)
                    dcl.set_cluster_sidecar(mdists, mrects, *side)
                    dcl.upload(mframes)

                    def host_chain(n, cap):
                        return [cluster_matches(m, mdists, mrects, *side) for m in dcl.collect(n, cap)]
                    cc = {}
                    for label, fn in (("device", lambda n, cap: dcl.collect_clusters(n, cap)), ("host", host_chain)):
                        run_pipelined(dcl, 2 * dcl.max_outstanding + 2, B, args.threshold, collect=fn)
                        out_c, dt_c, _ = timed(torch, lambda t0: run_pipelined(dcl, csteps, B, args.threshold, collect=fn))
                        cc[label] = {"value": B * csteps / dt_c, "unit": "frames/s", "ms_per_step": dt_c / csteps * 1e3,
                                     "clusters_per_frame": float(np.mean([len(o_[1] if label == "device" else o_[0]) for o_ in out_c]))}
                    cc["note"] = "std::sort + std::unique + rcd_voting + cluster_filter + cluster_scoring + IoU-NMS per frame; device = k_f2_finalize_cluster, only matches and clusters cross PCIe"
                    dcl.close()
                    mb["collect_clusters"] = cc
                    # addTemplate on rendered views, device trainer vs the one-core oracle trainer
                    try:
                        from linemod_pose_estimation_amd import NativeBank
                        tviews = [ms.training_view(chip, *views[i]) for i in range(0, 2652, 22)]
                        nbt = NativeBank.create(mbank.T, mbank.modalities)
                        nbt.add_template([tviews[0][0], tviews[0][1]], "warm", tviews[0][2], device=local_rank)
                        t_a = time.perf_counter()
                        for bgr_, depth_, mask_, _ in tviews:
                            nbt.add_template([bgr_, depth_], "obj", mask_, device=local_rank)
                        t_dev = time.perf_counter() - t_a
                        tr = {"views": len(tviews), "device_views_per_s": len(tviews) / t_dev, "ms_per_view": t_dev / len(tviews) * 1e3,
                              "note": "lmx_bank_add_template on 640x480 RGB-D training views rendered from the reference's mesh (per-pixel stages on the device, greedy selection on the host)"}
                        if not args.no_cpu_baseline:
                            from oracle import oracle as o_
                            odt = o_.OracleDetector(ms.empty_bank())
                            t_a = time.perf_counter()
                            for bgr_, depth_, mask_, _ in tviews[:40]:
                                odt.add_template([bgr_, depth_], "obj", mask_)
                            tr["cpu_baseline_views_per_s"] = 40 / (time.perf_counter() - t_a)
                        mb["trainer"] = tr
                    except Exception as e:
                        mb["trainer"] = {"error": str(e)[:200]}
                    extra["mesh_bank"] = mb
                    del mbank, mframes, distinct
                    # BASELINE configs[2] as it is worded: the two objects it names, both banks rendered from the reference's meshes
                    # (2 x 2652 templates, two classes), 1280x960 scenes holding rendered instances of both
                    try:
                        bank2, _ = ms.load_banks(("memoryChip2", "cpu_binary"))
                        d2 = [ms.make_scene(chip, views, 1280, 960, seed=7100 + f, n_instances=4, other_tri=cpu_mesh, n_other=4, other_class="cpu_binary",
                                            texture=args.texture)[0] for f in range(4)]
                        c3m = secondary_line(torch, Detector, bank2, [d2[f % len(d2)] for f in range(16)], 16, args.threshold, csteps, width=1280, height=960)
                        c3m["workload"] = ("BASELINE configs[2] with rendered banks: memoryChip2 + cpu_binary, 2 x 2652 templates from the reference's meshes, 1280x960, "
                                           "16 resident frames per step (4 distinct scenes: 4 chips + 4 cpus each)")
                        extra["config3_mesh_two_objects_1280x960"] = c3m
                        del bank2, d2
                    except Exception as e:
                        extra["config3_mesh_two_objects_1280x960"] = {"error": str(e)[:300]}
                except Exception as e:
                    extra["mesh_bank"] = {"error": str(e)[:300]}
                # the C++ device group on this one GPU: RCCL with one member (the exact call sequence of a multi-GPU group), and eight
                # members sharing the device (peer-copy collective) for the host-side cost of driving eight members
                try:
                    extra["group_1_member_rccl"] = group_line(torch, bank, frames, B, args.threshold, csteps, 1, "rccl")
                    g8 = group_line(torch, bank, frames, B, args.threshold, 24, 8, "peer_copy")
                    g8h = group_line(torch, bank, frames, B, args.threshold, 24, 8, "peer_copy", host_batches=[Detector.prepare_batch(b) for b in host_batches])
                    extra["group_8_members_one_gpu"] = {"resident": g8, "host_frames": g8h,
                                                        "note": "8 members share ONE device: the device runs 8 x the replicated pre-processing, so `value` is not a multi-GPU figure; "
                                                                "host_us_per_batch is what carries over to an 8-GPU node (DESIGN.md section 4)"}
                except Exception as e:
                    extra["group_1_member_rccl"] = {"error": str(e)[:300]}
            # the reference's own call pattern: ONE frame per call (the service node matches one camera frame per request,
            # ..._service.cpp:324-344).  Latency of lmx_match with a fresh pageable host frame, and of enqueue + collect on a resident one.
            try:
                d1 = Detector(bank, WIDTH, HEIGHT, device=local_rank, max_batch=1)
                singles = [[np.array(src, copy=True) for src in frames[i]] for i in range(min(B, 16))]

                def lat(fn, n=200, warm=20):
                    ts = []
                    for i in range(warm + n):
                        t_a = time.perf_counter()
                        fn(i)
                        ts.append(time.perf_counter() - t_a)
                    ts = np.asarray(ts[warm:]) * 1e6
                    return {"median": float(np.median(ts)), "p10": float(np.percentile(ts, 10)), "p90": float(np.percentile(ts, 90)), "n": n}
                # lmx_image descriptors built once per frame, as a C++ caller holding cv::Mat headers has them; `host_frame_marshalled_us` adds
                # the Python-side construction of the descriptors on every call
                prepared = [Detector.prepare_batch([f]) for f in singles]
                one = {"host_frame_us": lat(lambda i: d1.match_prepared(prepared[i % len(prepared)], args.threshold)),
                       "host_frame_marshalled_us": lat(lambda i: d1.match(singles[i % len(singles)], args.threshold))}
                d1.upload([singles[0]])

                def resident(i):
                    d1.enqueue(1, args.threshold)
                    d1.collect(1)
                one["resident_frame_us"] = lat(resident)
                one["note"] = "one 640x480 RGB-D frame per call against the 3000-template bank, through the ctypes binding; the CPU baseline needs cpu_baseline.ms_per_frame for the same call"
                d1.close()
                extra["single_frame_latency"] = one
            except Exception as e:
                extra["single_frame_latency"] = {"error": str(e)[:200]}
            line["extra"] = extra
        if dist_extra is not None:
            line["rccl_world"] = dist_extra["communicator"]["rccl_world"]
            if "host_frames" in dist_extra:
                line["host_frames"] = dist_extra.pop("host_frames")
            line["extra"] = dist_extra
        if not args.no_cpu_baseline and world == 1:  # the host baseline is timed at N=1 only (rank 0), as the contract asks
            line["cpu_baseline"] = cpu_baseline(bank, frames, args.threshold)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(bank, frames, args.threshold)
            end_to_end = line.get("host_frames", {}).get("value")
            line["speedup_vs_cpu_1core"] = (end_to_end or value) / line["cpu_baseline"]["value"]
            line["speedup_vs_cpu_1core_basis"] = "host_frames (fresh pageable host frames every step)" if end_to_end else "value (device-resident frames)"
            line["speedup_resident_vs_cpu_1core"] = value / line["cpu_baseline"]["value"]
            line["speedup_vs_cpu_all_cores"] = (end_to_end or value) / line["cpu_baseline_all_cores"]["value"]
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
