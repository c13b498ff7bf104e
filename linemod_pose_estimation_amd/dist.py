"""Multi-GPU plumbing for the template-sharded matcher.

One process per GPU.  The ranks form a frame_groups x template_shards grid (world = G * R, rank k = (g, r) = (k // R, k % R)):
rank (g, r) holds templates [r*N/R, (r+1)*N/R) of every class resident in its HBM (`Detector(..., shard_rank=r, shard_world=R)`)
and takes frames [g*n/G, (g+1)*n/G) of every batch of n frames.  G = 1 (the default): every rank pre-processes the same frames
(cheaper than moving linear memories over xGMI) and scores its template shard -- one frame's latency, BASELINE configs[3].
R = 1: every rank holds the whole bank (tens of MB at 50 000 templates) and takes its share of the frames, so nothing is
replicated -- a stream of frames, BASELINE configs[4].  Either way the only exchange is ONE all-gather per frame batch of a
fixed-capacity per-rank block {64-byte header with the record count, lmx_raw_match_t[K]} over torch.distributed (backend "nccl"
== RCCL on ROCm; "gloo" on CPU / for ranks sharing a GPU in tests).  The payload is tens of KB per rank: latency-bound, so it is
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
    synchronously inside `submit`.

    A rank that produced more records than `gather_capacity` is not an error: every rank reads the same gathered headers, so
    every rank takes the same decision in `finish` -- re-allocate the batch's blocks to fit, export the batch again from the
    records still held in the context's output slot (lmx_ctx_export_oldest_on) and repeat the all-gather (SURVEY 8e's two-phase
    fallback, as lmx_group_finish does it from C++).  Later batches use the grown capacity when their ring entry comes round.

    `result_ranks`: None = every rank merges (every rank returns the matches); a collection of ranks = only those copy the
    gathered blocks to the host and merge, the others return None from `finish` (a job whose consumer lives on rank 0)."""

    def __init__(self, bank, width, height, max_batch=1, gather_capacity=8192, max_candidates=0, group=None, overlap=True, result_ranks=None, frame_groups=1,
                 hipgraph=False):
        from .detector import Detector
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.G = max(1, int(frame_groups))
        if self.world % self.G != 0:
            raise ValueError("ShardedMatcher: %d ranks do not form %d frame groups" % (self.world, self.G))
        self.R = self.world // self.G
        self.fgroup, self.shard = self.rank // self.R, self.rank % self.R
        self.max_batch = max_batch                 # of the whole batch; this rank holds ceil(max_batch / G) frames
        self.n_uploaded = 0
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.on_device = (not dist.is_initialized()) or dist.get_backend(group) != "gloo"   # RCCL gathers device buffers in place
        if dist.is_initialized() and self.on_device:
            # RCCL sets itself up lazily, at the first collective of a communicator (its streams, channels, buffers).  When that happens AFTER
            # the detector's streams exist the matching kernels run 2.5 % slower for the life of the process (measured, one rank: 138.5 k
            # against 142.1 k frames/s, scripts/sharded_overhead_split.py); so the communicator does its first collective here, before the
            # context is created.
            warm = torch.zeros(256, dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(torch.empty(256 * self.world, dtype=torch.uint8, device=self.device), warm, group=group)
            torch.cuda.synchronize(self.device)
        self.det = Detector(bank, width, height, device=self.device.index, max_batch=(max_batch + self.G - 1) // self.G, max_candidates=max_candidates,
                            shard_rank=self.shard, shard_world=self.R, overlap=overlap, hipgraph=hipgraph)
        self.depth = self.det.max_outstanding
        self.capacity = gather_capacity          # for ring entries (re)allocated from now on
        self.wants_result = result_ranks is None or self.rank in result_ranks
        self.comm = torch.cuda.Stream(device=self.device)
        # the collective runs whenever a device-side process group exists, also for one rank, so that the exact RCCL call sequence
        # of the multi-GPU job is what the single-GPU tests and `bench.py --sharded` execute
        self.collective = dist.is_initialized() and self.on_device
        self.cap = [0] * self.depth               # per ring entry: the capacity its buffers are sized for
        self.send, self.recv, self.host = [None] * self.depth, [None] * self.depth, [None] * self.depth
        for k in range(self.depth):
            self._alloc_entry(k, gather_capacity)
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.mine = [0] * self.depth              # per ring entry: frames of the batch that are this rank's
        self.pending = []   # (buffer index, n_frames, blocks or None) oldest first
        self.head = 0
        self.regrows = 0

    def _alloc_entry(self, k, capacity):
        """Buffers of ring entry k for `capacity` records per rank (the entry is idle: its previous batch was finished)."""
        self.comm.synchronize()
        nbytes = block_bytes(capacity)
        self.send[k] = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.recv[k] = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device) if self.collective else self.send[k]
        self.host[k] = torch.empty(self.world * nbytes, dtype=torch.uint8).pin_memory()
        self.cap[k] = capacity

    def frames_of(self, n):
        """(first, count): this rank's frame group's slice of a batch of n frames."""
        first = self.fgroup * n // self.G
        return first, (self.fgroup + 1) * n // self.G - first

    def upload(self, frames):
        """The WHOLE batch (every rank is handed the same frames); the rank uploads its frame group's slice of it."""
        from .detector import PreparedBatch
        n = frames.n_frames if isinstance(frames, PreparedBatch) else len(frames)
        if n < 1 or n > self.max_batch:
            raise ValueError("ShardedMatcher.upload: %d frames outside [1, %d]" % (n, self.max_batch))
        first, count = self.frames_of(n)
        self.n_uploaded = n
        if count == 0:
            return        # a batch with fewer frames than frame groups leaves this rank idle
        if self.G == 1:
            self.det.upload(frames)
        else:
            self.det.upload(frames.slice(first, count) if isinstance(frames, PreparedBatch) else frames[first:first + count])

    def _exchange(self, k, oldest):
        """export (most recent enqueue, or the oldest outstanding one) -> all-gather -> host copy of entry k, on the communication stream.
        Returns the gathered blocks when the exchange had to be synchronous (gloo), else None (wait for ready[k])."""
        cap, nbytes = self.cap[k], block_bytes(self.cap[k])
        if not self.mine[k]:
            with torch.cuda.stream(self.comm):
                self.send[k][:GATHER_HEADER_BYTES].zero_()      # no frames of this batch fell to this rank: an empty block
        elif oldest:
            _lib.check(_lib.lib().lmx_ctx_export_oldest_on(self.det.h, self.send[k].data_ptr(), cap, self.comm.cuda_stream))
        else:
            # the communication stream waits (on the device) for this enqueue, then carries copy -> all-gather -> read-back
            self.det.export_raw_on(self.send[k].data_ptr(), cap, self.comm.cuda_stream)
        if self.on_device:
            with torch.cuda.stream(self.comm):
                if self.collective:
                    dist.all_gather_into_tensor(self.recv[k], self.send[k], group=self.group)
                # read-back by a copy kernel writing through the mapping of the pinned buffer, not by DMA; per rank only the header and
                # the records it counts.  Every rank reads the headers (they carry the regrow decision); a rank that does not merge
                # reads nothing else
                _lib.check(_lib.lib().lmx_stream_copy_blocks(self.host[k].data_ptr(), self.recv[k].data_ptr(), self.world, nbytes, cap if self.wants_result else 0,
                                                             self.comm.cuda_stream))
                self.ready[k].record(self.comm)
            return None
        self.comm.synchronize()
        return allgather_blocks(self.send[k].cpu(), self.group).contiguous().numpy().reshape(-1)

    def submit(self, n_frames, threshold):
        if len(self.pending) >= self.depth:
            raise RuntimeError("ShardedMatcher: %d batches are already in flight; finish one first" % self.depth)
        k = self.head
        self.head = (k + 1) % self.depth
        if self.cap[k] < self.capacity:     # an earlier batch made the blocks grow
            self._alloc_entry(k, self.capacity)
        if self.G > 1 and n_frames != self.n_uploaded:
            raise ValueError("ShardedMatcher.submit: with frame groups a batch is the whole upload (%d frames), not %d" % (self.n_uploaded, n_frames))
        self.mine[k] = self.frames_of(n_frames)[1] if self.G > 1 else n_frames
        if self.mine[k]:
            self.det.enqueue(self.mine[k], threshold)
        self.pending.append((k, n_frames, self._exchange(k, False)))

    def finish(self):
        """Per-frame matches of the oldest batch in flight (None on a rank outside `result_ranks`)."""
        k, n_frames, blocks = self.pending.pop(0)
        try:
            if blocks is None:
                self.ready[k].synchronize()
                blocks = self.host[k].numpy()
            nbytes = block_bytes(self.cap[k])
            need = max(int(blocks[r * nbytes + 4:r * nbytes + 8].view(np.uint32)[0]) for r in range(self.world))
            if need > self.cap[k]:
                # the same headers on every rank -> the same decision on every rank: the repeated all-gather lines up
                grown = max(self.cap[k], 1)
                while grown < need:
                    grown *= 2
                self.capacity = max(self.capacity, grown)
                self._alloc_entry(k, grown)
                self.regrows += 1
                blocks = self._exchange(k, True)
                if blocks is None:
                    self.ready[k].synchronize()
                    blocks = self.host[k].numpy()
            out = merge_gathered(blocks, self.world, block_bytes(self.cap[k]), self.cap[k], n_frames, frame_groups=self.G) if self.wants_result else None
        finally:
            if self.mine[k]:
                self.det.release()   # the enqueue behind this batch has finished (the exchange waited for it): frees its slot either way
        return out

    def step(self, n_frames, threshold):
        """submit + finish of one batch.  Returns per-frame matches."""
        self.submit(n_frames, threshold)
        return self.finish()


class DeviceGroup:
    """lmx_group_* (csrc/lmx_group.cpp) from Python: ONE process drives several members -- the GPUs of a node, or, with
    collective="peer_copy", several members sharing a device (tests, single-GPU rehearsal of the group's host side).  The data
    path is the C++ one the reference's caller would use; this class only marshals frames and results."""

    def __init__(self, bank, width, height, n_members, devices=None, max_batch=1, gather_capacity=8192, max_candidates=0, collective="rccl",
                 overlap=True, hipgraph=False, frame_groups=1):
        import ctypes as C
        from .detector import NativeBank
        from .bank import TemplateBank
        self.native_bank = NativeBank.from_bank(bank) if isinstance(bank, TemplateBank) else bank
        devs = list(devices) if devices is not None else list(range(n_members))
        self._devs = (C.c_int32 * n_members)(*devs)
        desc = _lib.GroupDesc(n_members, self._devs, width, height, max_batch, max_candidates, gather_capacity, (1 if hipgraph else 0) | (2 if overlap else 0), None, 0, 0, 0,
                              {"rccl": 0, "peer_copy": 1}[collective], frame_groups)
        self.h = C.c_void_p()
        _lib.check(_lib.lib().lmx_group_create(self.native_bank.h, C.byref(desc), C.byref(self.h)))
        self.depth = int(_lib.lib().lmx_group_depth(self.h))
        self.size = int(_lib.lib().lmx_group_size(self.h))
        self.collective = _lib.lib().lmx_group_collective_name(self.h).decode()
        self.frame_groups = int(_lib.lib().lmx_group_frame_groups(self.h))

    def upload(self, frames):
        from .detector import PreparedBatch, _images
        if isinstance(frames, PreparedBatch):
            _lib.check(_lib.lib().lmx_group_upload(self.h, frames.n_frames, frames.imgs, frames.n_sources))
            return
        imgs, keep = _images(frames)
        _lib.check(_lib.lib().lmx_group_upload(self.h, len(frames), imgs, len(frames[0])))
        del keep

    def submit(self, n_frames, threshold):
        import ctypes as C
        _lib.check(_lib.lib().lmx_group_submit(self.h, n_frames, C.c_float(threshold), None, 0))

    def finish(self, n_frames, cap=4096):
        import ctypes as C
        from .detector import MATCH_DTYPE
        if getattr(self, "_out", None) is None or self._out.shape != (n_frames, cap):
            self._out = np.zeros((n_frames, cap), MATCH_DTYPE)
        n_out = (C.c_size_t * n_frames)()
        _lib.check(_lib.lib().lmx_group_finish(self.h, n_frames, self._out.ctypes.data, cap, n_out))
        return [self._out[f, :n_out[f]].copy() for f in range(n_frames)]

    def gather_capacity(self):
        return int(_lib.lib().lmx_group_gather_capacity(self.h))

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().lmx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
