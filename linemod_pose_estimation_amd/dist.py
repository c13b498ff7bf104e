"""Multi-GPU plumbing for the template-sharded matcher.

One process per GPU; rank r holds templates [r*N/R, (r+1)*N/R) of every class resident in its HBM
(`Detector(..., shard_rank=r, shard_world=R)`), every rank pre-processes the same frames (cheaper than moving
linear memories over xGMI), and the only exchange is ONE all-gather per frame batch of a fixed-capacity per-rank
block {64-byte header with the record count, lmx_raw_match_t[K]} over torch.distributed (backend "nccl" == RCCL on
ROCm; "gloo" on CPU / for ranks sharing a GPU in tests).  The payload is tens of KB per rank: latency-bound, so it is
batched over all frames of a step.  The host then merges the gathered records per frame with the same
std::sort/std::unique as a single GPU would (`lmx_merge_gathered`): records sorted by order_key reproduce upstream
insertion order, so the result is identical for any shard count.
"""
import numpy as np
import torch
import torch.distributed as dist

from .detector import GATHER_HEADER_BYTES, RAW_MATCH_DTYPE, merge_gathered

RECORD_BYTES = RAW_MATCH_DTYPE.itemsize  # 32


def block_bytes(capacity):
    return GATHER_HEADER_BYTES + capacity * RECORD_BYTES


def make_block(records, capacity):
    """Host-side builder of one rank's gather block (tests; the GPU path writes it with lmx_ctx_export_raw)."""
    blk = np.zeros(block_bytes(capacity), np.uint8)
    blk[:8].view(np.uint32)[1] = len(records)
    n = min(len(records), capacity)
    blk[GATHER_HEADER_BYTES:GATHER_HEADER_BYTES + n * RECORD_BYTES] = np.ascontiguousarray(records[:n], RAW_MATCH_DTYPE).view(np.uint8)
    return blk


def allgather_blocks(local_block: torch.Tensor, group=None):
    """local_block: uint8 [block_bytes] on any device -> uint8 [world, block_bytes] (same device, or CPU under gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_block.reshape(1, -1)
    if dist.get_backend(group) == "gloo" and local_block.is_cuda:
        local_block = local_block.cpu()  # gloo gathers host tensors; RCCL gathers the device buffers in place
    out = torch.empty(world * local_block.numel(), dtype=torch.uint8, device=local_block.device)
    dist.all_gather_into_tensor(out, local_block.reshape(-1), group=group)
    return out.view(world, -1)


class ShardedMatcher:
    """Template-sharded detector for one rank of a torch.distributed job."""

    def __init__(self, bank, width, height, max_batch=1, gather_capacity=8192, max_candidates=0, group=None):
        from .detector import Detector
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream().cuda_stream
        self.det = Detector(bank, width, height, device=self.device.index, max_batch=max_batch, max_candidates=max_candidates,
                            shard_rank=self.rank, shard_world=self.world, stream=stream)
        self.capacity = gather_capacity
        self.block = block_bytes(gather_capacity)
        self.send = torch.zeros(self.block, dtype=torch.uint8, device=self.device)
        self.host = torch.empty(self.world * self.block, dtype=torch.uint8).pin_memory()

    def upload(self, frames):
        self.det.upload(frames)

    def step(self, n_frames, threshold):
        """enqueue on this rank's shard -> export the gather block -> all-gather -> host merge.  Returns per-frame matches."""
        self.det.enqueue(n_frames, threshold)
        self.det.export_raw(self.send.data_ptr(), self.capacity)   # one D2D copy on the shared stream
        gathered = allgather_blocks(self.send, self.group)
        if gathered.is_cuda:
            self.host.view(self.world, -1).copy_(gathered, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            blocks = self.host.numpy()
        else:
            blocks = gathered.contiguous().numpy().reshape(-1)
        out = merge_gathered(blocks, self.world, self.block, self.capacity, n_frames)
        self.det.sync()
        return out
