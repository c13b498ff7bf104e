"""Multi-GPU plumbing for the template-sharded matcher.

One process per GPU; rank r holds templates [r*N/R, (r+1)*N/R) of every class resident in its HBM
(`Detector(..., shard_rank=r, shard_world=R)`), every rank pre-processes the same frames (cheaper than moving
linear memories over xGMI), and the only exchange is ONE all-gather per frame batch of fixed-capacity per-rank
buffers {count, lmx_raw_match_t[K]} over torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" on CPU for
tests).  The payload is tens of KB per rank: latency-bound, so it is batched over all frames of a step.
The host then merges the gathered records per frame with the same std::sort/std::unique as a single GPU would
(`lmx_merge_raw`): concatenation in order_key order equals upstream insertion order, so the result is
identical for any shard count.
"""
import numpy as np
import torch
import torch.distributed as dist

from .detector import RAW_MATCH_DTYPE, merge_raw

RECORD_BYTES = RAW_MATCH_DTYPE.itemsize  # 32


def allgather_records(local_records: torch.Tensor, local_count: torch.Tensor, group=None):
    """local_records: uint8 [K*32] (device or CPU tensor), local_count: int32 [1] on the same device.
    Returns (records uint8 [world, K*32], counts int32 [world]) on that device."""
    world = dist.get_world_size(group)
    n = local_records.numel()
    rec = torch.empty(world * n, dtype=local_records.dtype, device=local_records.device)
    cnt = torch.empty(world * local_count.numel(), dtype=local_count.dtype, device=local_count.device)
    dist.all_gather_into_tensor(rec, local_records.reshape(-1), group=group)
    dist.all_gather_into_tensor(cnt, local_count.reshape(-1), group=group)
    return rec.view(world, n), cnt.view(world, -1)[:, 0]


def merge_gathered(records: torch.Tensor, counts: torch.Tensor, n_frames: int, capacity: int):
    """Host side: gathered buffers -> list (per frame) of final matches in upstream output order."""
    counts_h = counts.cpu().numpy()
    if (counts_h > capacity).any():
        raise OverflowError("a rank produced %d records > all-gather capacity %d" % (int(counts_h.max()), capacity))
    rec_h = records.cpu().numpy().view(np.uint8).reshape(records.shape[0], -1)
    parts = [rec_h[r, : int(counts_h[r]) * RECORD_BYTES].view(RAW_MATCH_DTYPE) for r in range(rec_h.shape[0])]
    allrec = np.concatenate(parts) if parts else np.zeros(0, RAW_MATCH_DTYPE)
    return [merge_raw(allrec[allrec["frame"] == f]) for f in range(n_frames)]


class ShardedMatcher:
    """Template-sharded detector for one rank of a torch.distributed job."""

    def __init__(self, bank, width, height, max_batch=1, gather_capacity=4096, max_candidates=0, group=None):
        from .detector import Detector
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream().cuda_stream
        self.det = Detector(bank, width, height, device=self.device.index, max_batch=max_batch, max_candidates=max_candidates,
                            shard_rank=self.rank, shard_world=self.world, stream=stream)
        self.capacity = gather_capacity
        self.records = torch.zeros(gather_capacity * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int32, device=self.device)

    def upload(self, frames):
        self.det.upload(frames)

    def step(self, n_frames, threshold):
        """enqueue on this rank's shard -> export raw records -> all-gather -> host merge.  Returns per-frame matches."""
        self.det.enqueue(n_frames, threshold)
        # device-to-device copies on the shared stream into the fixed-capacity all-gather send buffers
        self.det.export_raw(self.records.data_ptr(), self.capacity, self.count.data_ptr())
        if self.world > 1:
            rec, cnt = allgather_records(self.records, self.count, self.group)
        else:
            rec, cnt = self.records[None], self.count
        out = merge_gathered(rec, cnt, n_frames, self.capacity)  # .cpu() synchronises the stream
        self.det.sync()
        return out
