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

from . import _lib
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
    """Template-sharded detector for one rank of a torch.distributed job.

    `submit` queues one batch (kernels on the context's device lanes, then export -> all-gather -> copy to pinned host
    memory on a separate communication stream) and returns at once; `finish` waits for the oldest submitted batch and merges
    it on the host.  Up to `depth` batches may be in flight, so the exchange and the host merge of batch i overlap the kernels
    of the following batches.  `step` = submit + finish.  Under gloo (CPU tests, ranks sharing one GPU) the exchange is done
    synchronously inside `submit`."""

    def __init__(self, bank, width, height, max_batch=1, gather_capacity=8192, max_candidates=0, group=None, overlap=True):
        from .detector import Detector
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.on_device = (not dist.is_initialized()) or dist.get_backend(group) != "gloo"   # RCCL gathers device buffers in place
        self.det = Detector(bank, width, height, device=self.device.index, max_batch=max_batch, max_candidates=max_candidates,
                            shard_rank=self.rank, shard_world=self.world, overlap=overlap)
        self.depth = self.det.max_outstanding
        self.capacity = gather_capacity
        self.block = block_bytes(gather_capacity)
        self.comm = torch.cuda.Stream(device=self.device)
        self.send = [torch.zeros(self.block, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        # the collective runs whenever a device-side process group exists, also for one rank, so that the exact RCCL call sequence
        # of the multi-GPU job is what the single-GPU tests and `bench.py --sharded` execute
        self.collective = dist.is_initialized() and self.on_device
        self.recv = [torch.empty(self.world * self.block, dtype=torch.uint8, device=self.device) for _ in range(self.depth)] if self.collective else self.send
        self.host = [torch.empty(self.world * self.block, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.pending = []   # (buffer index, n_frames, blocks or None) oldest first
        self.head = 0

    def upload(self, frames):
        self.det.upload(frames)

    def submit(self, n_frames, threshold):
        if len(self.pending) >= self.depth:
            raise RuntimeError("ShardedMatcher: %d batches are already in flight; finish one first" % self.depth)
        k = self.head
        self.head = (k + 1) % self.depth
        self.det.enqueue(n_frames, threshold)
        # the communication stream waits (on the device) for this enqueue, then carries copy -> all-gather -> read-back
        self.det.export_raw_on(self.send[k].data_ptr(), self.capacity, self.comm.cuda_stream)
        blocks = None
        if self.on_device:
            with torch.cuda.stream(self.comm):
                if self.collective:
                    dist.all_gather_into_tensor(self.recv[k], self.send[k], group=self.group)
                # read-back by a copy kernel writing through the mapping of the pinned buffer, not by DMA; per rank only the
                # header and the records it counts
                _lib.check(_lib.lib().lmx_stream_copy_blocks(self.host[k].data_ptr(), self.recv[k].data_ptr(), self.world, self.block, self.capacity,
                                                             self.comm.cuda_stream))
                self.ready[k].record(self.comm)
        else:
            self.comm.synchronize()
            blocks = allgather_blocks(self.send[k].cpu(), self.group).contiguous().numpy().reshape(-1)
        self.pending.append((k, n_frames, blocks))

    def finish(self):
        """Per-frame matches of the oldest batch in flight."""
        k, n_frames, blocks = self.pending.pop(0)
        if blocks is None:
            self.ready[k].synchronize()
            blocks = self.host[k].numpy()
        try:
            out = merge_gathered(blocks, self.world, self.block, self.capacity, n_frames)
        finally:
            self.det.release()   # the enqueue behind this batch has finished (the exchange waited for it): frees its slot either way
        return out

    def step(self, n_frames, threshold):
        """submit + finish of one batch.  Returns per-frame matches."""
        self.submit(n_frames, threshold)
        return self.finish()
