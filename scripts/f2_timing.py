"""Cost of the device-side consumer chain (k_f2_finalize_cluster: std::sort + std::unique + rcd_voting / filter / scoring / IoU-NMS,
csrc/lmx_f2.hip) at different record counts per frame, on the mesh-rendered bank with its renderer-params side-car.  Prints, per
threshold, raw records / final matches / clusters per frame and the wall time of collect_clusters vs collect + host lmx_cluster_matches.
Run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import Detector, meshsynth as ms
from linemod_pose_estimation_amd.detector import cluster_matches
bank, rects, dists, views_idx = ms.load_bank("memoryChip2")
chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = [ms.make_scene(chip, views, seed=7000 + f, n_instances=3, other_tri=cpu, n_other=2)[0] for f in range(B)]
det = Detector(bank, 640, 480, max_batch=B, max_candidates=1 << 17)
det.set_cluster_sidecar(dists, rects, 8, ms.ENSENSO["radius_min"], ms.ENSENSO["radius_step"], 2)
det.upload(frames)
for thr in (92.0, 88.0, 85.0, 83.0, 81.0):
    for rep in range(3):
        det.enqueue(B, thr)
        t0 = time.perf_counter()
        out = det.collect_clusters(B)
        t_dev = time.perf_counter() - t0
    raw = det.stats()["raw_matches"] / B
    for rep in range(3):
        det.enqueue(B, thr)
        t0 = time.perf_counter()
        ms_ = det.collect(B)
        host = [cluster_matches(m, dists, rects, 8, ms.ENSENSO["radius_min"], ms.ENSENSO["radius_step"], 2) for m in ms_]
        t_host = time.perf_counter() - t0
    print("thr %.0f: raw records/frame %.0f, final matches/frame %.0f, clusters/frame %.1f | collect_clusters (device chain) %.2f ms, collect + host chain %.2f ms, %d frames"
          % (thr, raw, np.mean([len(o[0]) for o in out]), np.mean([len(o[1]) for o in out]), t_dev * 1e3, t_host * 1e3, B), flush=True)
