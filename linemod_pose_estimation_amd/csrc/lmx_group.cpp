// Multi-GPU matching from the C++ side: lmx_group_* of include/lmx.h.
// The caller of the hot path is C++ (rgbdDetector::linemod_detection, /root/reference/src/rgbdDetector.cpp:31-34), so sharding
// must not need Python: a group owns one device context per GPU.  The members form a G x R grid (frame groups x template shards):
// member k = (g, r) = (k / R, k % R) holds templates [r*N/R, (r+1)*N/R) of every class and takes frames [g*n/G, (g+1)*n/G) of a
// batch of n frames -- G = 1: every member pre-processes the same frames and scores its template shard (one frame's latency);
// R = 1: every member holds the whole bank and takes its share of the frames (a stream of frames: nothing is replicated).  The
// exchange is ONE all-gather of fixed-capacity per-member blocks {64-byte header, lmx_raw_match_t[K]} per batch; the host merges
// rank 0's copy with the same std::sort / std::unique a single GPU runs (lmx_merge_gathered_groups), so the result equals the
// 1-GPU result for any G x R.
//   single process, all GPUs of the node : ncclCommInitAll over the chosen devices (the C++ node process)
//   one process per GPU                   : ncclCommInitRank with an id from lmx_group_unique_id (torchrun-style launchers)
// The exchange sits behind a small ops table (`Collective`): RCCL over xGMI (dlopen'ed on first use, liblmx.so itself does not
// depend on it) or, single process only, device-to-device block copies between the members' buffers ("peer_copy"), which also
// works when several members share one device -- that is how groups of 2, 3 and 8 members are tested on a one-GPU box.
// Pipelining: a ring of `depth` batches (send/receive blocks per member, one pinned host block set); submit() queues a batch on
// every member and returns, finish() completes the oldest.  Host frames are staged once and fanned out: one non-temporal copy
// into a pinned staging area, then one DMA per member and modality from that same area.
// Overflow of the gather block is not an error: the header carries every rank's record count, and when one exceeds the
// block's capacity the batch's blocks are re-allocated to fit and its exchange is repeated from the records still held in the
// contexts' output slots (two-phase "counts first" form of SURVEY 8e, paid only when it is needed).

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "lmx_internal.hpp"

namespace {

// the slice of rccl.h this file uses (same ABI; the header is not required at build time of a caller)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1 };

struct Rccl {
  void* lib = nullptr;
  std::string load_error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

Rccl* rccl() {
  std::call_once(g_rccl_once, []() {
    Rccl& r = g_rccl;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      const char* e = dlerror();   // one call: a second dlerror() returns NULL
      r.load_error = e ? e : "dlopen failed";
    }
    if (!r.lib) return;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(r.lib, name); if (!p) { ok = false; r.load_error = std::string("symbol missing: ") + name; } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
  });
  return g_rccl.lib ? &g_rccl : nullptr;
}

#define G_HIP(expr)                                                                                                     \
  do {                                                                                                                  \
    hipError_t e_ = (expr);                                                                                             \
    if (e_ != hipSuccess) { lmx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return LMX_ERR_HIP; } \
  } while (0)
#define G_NCCL(expr)                                                                                                    \
  do {                                                                                                                  \
    ncclResult_t r_ = (expr);                                                                                           \
    if (r_ != 0) { lmx::set_error("%s failed: %s", #expr, rccl()->GetErrorString(r_)); return LMX_ERR_HIP; }            \
  } while (0)

struct Member {          // one GPU (or one share of a GPU) of this process
  int device = 0;
  int rank = 0;            // position in the all-gather = fgroup * R + shard
  int fgroup = 0, shard = 0;
  lmx_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  std::vector<uint8_t*> d_send, d_recv;   // per ring entry
  std::vector<hipEvent_t> sent, pulled;   // peer_copy: the entry's block is exported / every block of the entry has been pulled
  std::vector<char> pulled_recorded;
};

struct RingEntry {       // one batch in flight
  size_t capacity = 0;   // records per rank block its buffers are sized for
  uint8_t* h_blocks = nullptr;   // pinned [world][block_bytes]: member 0's gathered view
  hipEvent_t ready = nullptr;    // recorded on member 0's communication stream behind the copy into h_blocks
  int n_frames = 0;
  std::vector<char> enqueued;    // per local member: it had frames in this batch (a batch of fewer frames than frame groups leaves groups idle)
};

struct Collective {
  const char* name;
  lmx_status (*init)(lmx_group*);
  lmx_status (*warm_up)(lmx_group*);             // one throw-away exchange right after init (null: nothing to set up lazily)
  lmx_status (*all_gather)(lmx_group*, int k);   // queue on every member's communication stream: d_send[k] of all ranks -> d_recv[k]
  void (*destroy)(lmx_group*);
};

}  // namespace

struct lmx_group {
  const lmx_bank* bank = nullptr;
  lmx_group_desc desc{};
  int world = 1;
  int G = 1, R = 1;              // frame groups x template shards = world
  int n_uploaded = 0;            // frames of the most recent upload (the whole batch, before it was dealt to the frame groups)
  bool multi_process = false;
  // frames [first, first + count) of a batch of n frames belong to frame group fg
  void frames_of(int fg, int n, int* first, int* count) const {
    *first = (int)((long)fg * n / G);
    *count = (int)((long)(fg + 1) * n / G) - *first;
  }
  const Collective* coll = nullptr;
  std::vector<Member> members;   // all ranks (single process) or this process's one rank
  std::vector<int> devlist;
  size_t capacity = 0;           // records per rank block for newly (re)allocated ring entries
  std::vector<RingEntry> ring;
  int depth = 2, head = 0, in_flight = 0;
  std::vector<uint8_t*> h_stage; // single process, several members: one pinned staging area per frame set (lock step with the members)
  bool uploaded = false;
  std::unique_ptr<lmx::CopyPool> pool;   // drives the members in parallel and stages host frames
  // LMX_GROUP_TRACE=1: host time per phase (calling thread, seconds), printed by lmx_group_destroy
  enum { T_BEGIN, T_STAGE, T_DMA, T_ENQUEUE, T_EXPORT, T_GATHER, T_COPY, T_WAIT, T_MERGE, T_RELEASE, T_COUNT };
  bool trace = false;
  double t_acc[T_COUNT] = {};
  long t_n[T_COUNT] = {};
  static size_t block_bytes(size_t capacity) { return LMX_GATHER_HEADER_BYTES + capacity * sizeof(lmx_raw_match_t); }
};

namespace {

struct Phase {   // scoped host timer of one phase of upload / submit / finish
  lmx_group* g; int id; std::chrono::steady_clock::time_point t0;
  Phase(lmx_group* g_, int id_) : g(g_), id(id_) { if (g->trace) t0 = std::chrono::steady_clock::now(); }
  ~Phase() { if (g->trace) { g->t_acc[id] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g->t_n[id] += 1; } }
};

// fn(member index) on the group's host threads; the first failure's status and message reach the calling thread
lmx_status for_members(lmx_group* g, const std::function<lmx_status(int)>& fn) {
  const int n = (int)g->members.size();
  std::vector<lmx_status> st((size_t)n, LMX_OK);
  std::vector<std::string> msg((size_t)n);
  auto run = [&](int i) {
    st[(size_t)i] = fn(i);
    if (st[(size_t)i] != LMX_OK) msg[(size_t)i] = lmx_last_error();   // thread-local on the worker
  };
  if (g->pool && n > 1) g->pool->parallel_for(n, run);
  else
    for (int i = 0; i < n; ++i) run(i);
  for (int i = 0; i < n; ++i)
    if (st[(size_t)i] != LMX_OK) { lmx::set_error("%s", msg[(size_t)i].c_str()); return st[(size_t)i]; }
  return LMX_OK;
}

void free_entry(lmx_group* g, int k) {
  for (Member& m : g->members) {
    if (m.d_send.size() <= (size_t)k) continue;   // a group that failed half-way through its construction
    (void)hipSetDevice(m.device);
    if (m.d_send[(size_t)k]) (void)hipFree(m.d_send[(size_t)k]);
    if (m.d_recv[(size_t)k]) (void)hipFree(m.d_recv[(size_t)k]);
    m.d_send[(size_t)k] = m.d_recv[(size_t)k] = nullptr;
  }
  RingEntry& e = g->ring[(size_t)k];
  if (e.h_blocks) { (void)hipHostFree(e.h_blocks); e.h_blocks = nullptr; }
  e.capacity = 0;
}

// (re)allocates ring entry k for `capacity` records per rank.  The entry must not be in flight on any member.
lmx_status alloc_entry(lmx_group* g, int k, size_t capacity) {
  for (Member& m : g->members) {   // a member other than rank 0 may still be inside the entry's previous all-gather
    G_HIP(hipSetDevice(m.device));
    G_HIP(hipStreamSynchronize(m.comm_stream));
  }
  free_entry(g, k);
  const size_t bb = lmx_group::block_bytes(capacity);
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    G_HIP(hipMalloc((void**)&m.d_send[(size_t)k], bb));
    G_HIP(hipMalloc((void**)&m.d_recv[(size_t)k], bb * (size_t)g->world));
    G_HIP(hipMemset(m.d_send[(size_t)k], 0, bb));
    m.pulled_recorded[(size_t)k] = 0;
  }
  G_HIP(hipSetDevice(g->members[0].device));
  G_HIP(hipHostMalloc((void**)&g->ring[(size_t)k].h_blocks, bb * (size_t)g->world, hipHostMallocMapped | hipHostMallocPortable));
  g->ring[(size_t)k].capacity = capacity;
  return LMX_OK;
}

// ---- collective: RCCL ------------------------------------------------------------------------------------------------------
lmx_status rccl_init(lmx_group* g) {
  Rccl* R = rccl();
  if (!R) { lmx::set_error("librccl.so could not be loaded (needed for the all-gather of a device group): %s", g_rccl.load_error.c_str()); return LMX_ERR_NOT_FOUND; }
  if (g->multi_process) {
    ncclUniqueId id;
    std::memcpy(&id, g->desc.unique_id, sizeof(id));
    G_HIP(hipSetDevice(g->members[0].device));
    G_NCCL(R->CommInitRank(&g->members[0].comm, g->world, id, g->desc.rank));
    return LMX_OK;
  }
  for (size_t i = 0; i < g->devlist.size(); ++i)
    for (size_t j = 0; j < i; ++j)
      if (g->devlist[i] == g->devlist[j]) {
        lmx::set_error("lmx_group_create: device %d appears twice; RCCL needs one device per rank (members that share a device need LMX_GROUP_COLLECTIVE_PEER_COPY)", g->devlist[i]);
        return LMX_ERR_INVALID_ARG;
      }
  std::vector<ncclComm_t> comms(g->members.size());
  G_NCCL(R->CommInitAll(comms.data(), (int)g->members.size(), g->devlist.data()));
  for (size_t i = 0; i < g->members.size(); ++i) g->members[i].comm = comms[i];
  return LMX_OK;
}
lmx_status rccl_all_gather(lmx_group* g, int k) {
  Rccl* R = rccl();
  const size_t bb = lmx_group::block_bytes(g->ring[(size_t)k].capacity);
  G_NCCL(R->GroupStart());
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    G_NCCL(R->AllGather(m.d_send[(size_t)k], m.d_recv[(size_t)k], bb, ncclUint8, m.comm, m.comm_stream));
  }
  G_NCCL(R->GroupEnd());
  return LMX_OK;
}
lmx_status rccl_warm_up(lmx_group* g) {
  Rccl* R = rccl();
  const size_t n = 256;
  std::vector<void*> send(g->members.size(), nullptr), recv(g->members.size(), nullptr);
  lmx_status st = LMX_OK;
  for (size_t i = 0; i < g->members.size() && st == LMX_OK; ++i) {
    if (hipSetDevice(g->members[i].device) != hipSuccess || hipMalloc(&send[i], n) != hipSuccess || hipMalloc(&recv[i], n * (size_t)g->world) != hipSuccess ||
        hipMemset(send[i], 0, n) != hipSuccess) { lmx::set_error("lmx_group_create: warm-up buffers: %s", hipGetErrorString(hipGetLastError())); st = LMX_ERR_HIP; }
  }
  if (st == LMX_OK) {
    ncclResult_t r = R->GroupStart();
    for (size_t i = 0; i < g->members.size() && r == 0; ++i) {
      (void)hipSetDevice(g->members[i].device);
      r = R->AllGather(send[i], recv[i], n, ncclUint8, g->members[i].comm, nullptr);   // the null stream: no stream of the group exists yet
    }
    const ncclResult_t r2 = R->GroupEnd();
    if (r != 0 || r2 != 0) { lmx::set_error("lmx_group_create: warm-up all-gather failed: %s", R->GetErrorString(r != 0 ? r : r2)); st = LMX_ERR_HIP; }
  }
  for (size_t i = 0; i < g->members.size(); ++i) {
    (void)hipSetDevice(g->members[i].device);
    if (st == LMX_OK && hipDeviceSynchronize() != hipSuccess) { lmx::set_error("lmx_group_create: warm-up all-gather did not complete"); st = LMX_ERR_HIP; }
    if (send[i]) (void)hipFree(send[i]);
    if (recv[i]) (void)hipFree(recv[i]);
  }
  return st;
}
void rccl_destroy(lmx_group* g) {
  Rccl* R = rccl();
  for (Member& m : g->members)
    if (m.comm && R) { (void)hipSetDevice(m.device); (void)R->CommDestroy(m.comm); m.comm = nullptr; }
}

// ---- collective: device-to-device block copies (single process) --------------------------------------------------------------
// Member j pulls every member i's send block of the entry into slot rank_i of its own receive buffer, on its own communication
// stream, behind i's `sent` event; i's next export into the same send block waits for every member's `pulled` event of the
// entry (submit does that).  Same device or peer access: a copy kernel that moves the header and the counted records only;
// otherwise hipMemcpyPeerAsync of the whole block.
lmx_status peer_init(lmx_group* g) {
  if (g->multi_process) { lmx::set_error("lmx_group_create: the peer-copy collective exists within one process only; ranks in different processes exchange over RCCL"); return LMX_ERR_INVALID_ARG; }
  for (Member& a : g->members)
    for (Member& b : g->members)
      if (a.device != b.device) {
        int can = 0;
        G_HIP(hipSetDevice(a.device));
        if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
          const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { lmx::set_error("hipDeviceEnablePeerAccess(%d -> %d) failed: %s", a.device, b.device, hipGetErrorString(e)); return LMX_ERR_HIP; }
          (void)hipGetLastError();
        }
      }
  return LMX_OK;
}
lmx_status peer_all_gather(lmx_group* g, int k) {
  const size_t cap = g->ring[(size_t)k].capacity, bb = lmx_group::block_bytes(cap);
  // phase 1: every member marks its export; phase 2 (all marks are recorded by then): every member waits for the others' marks and
  // pulls all blocks with ONE kernel (source pointers by value), or block by block with hipMemcpyPeerAsync where a source device is
  // not peer-accessible
  lmx_status st = for_members(g, [&](int i) -> lmx_status {
    Member& m = g->members[(size_t)i];
    G_HIP(hipSetDevice(m.device));
    G_HIP(hipEventRecord(m.sent[(size_t)k], m.comm_stream));   // behind the member's export into d_send[k]
    return LMX_OK;
  });
  if (st != LMX_OK) return st;
  return for_members(g, [&](int i) -> lmx_status {
    Member& dst = g->members[(size_t)i];
    G_HIP(hipSetDevice(dst.device));
    bool all_visible = true;
    for (Member& src : g->members) {
      if (&src != &dst) G_HIP(hipStreamWaitEvent(dst.comm_stream, src.sent[(size_t)k], 0));
      int can = 1;
      if (src.device != dst.device && hipDeviceCanAccessPeer(&can, dst.device, src.device) != hipSuccess) can = 0;
      all_visible = all_visible && can;
    }
    if (all_visible) {
      for (size_t base = 0; base < g->members.size(); base += lmx::kPullMax) {
        lmx::PullSources ps{};
        const int n = (int)std::min<size_t>(lmx::kPullMax, g->members.size() - base);
        for (int j = 0; j < n; ++j) ps.src[j] = g->members[base + (size_t)j].d_send[(size_t)k];   // members are in rank order
        lmx::launch_pull_blocks(dst.comm_stream, dst.d_recv[(size_t)k] + base * bb, ps, n, bb, (uint32_t)std::min<size_t>(cap, 0xffffffffu));
      }
      G_HIP(hipGetLastError());
    } else {
      for (Member& src : g->members)
        G_HIP(hipMemcpyPeerAsync(dst.d_recv[(size_t)k] + (size_t)src.rank * bb, dst.device, src.d_send[(size_t)k], src.device, bb, dst.comm_stream));
    }
    G_HIP(hipEventRecord(dst.pulled[(size_t)k], dst.comm_stream));
    dst.pulled_recorded[(size_t)k] = 1;
    return LMX_OK;
  });
}
void peer_destroy(lmx_group*) {}

const Collective kRccl = {"rccl", rccl_init, rccl_warm_up, rccl_all_gather, rccl_destroy};
const Collective kPeerCopy = {"peer_copy", peer_init, nullptr, peer_all_gather, peer_destroy};

// queue "export -> all-gather -> rank 0's view to pinned host memory" of ring entry k; `oldest`: re-export the oldest outstanding
// enqueue of every member (regrow path) instead of the most recent one
lmx_status queue_exchange(lmx_group* g, int k, bool oldest) {
  RingEntry& e = g->ring[(size_t)k];
  lmx_status st;
  {
    Phase ph(g, lmx_group::T_EXPORT);
    st = for_members(g, [&](int i) -> lmx_status {
    Member& m = g->members[(size_t)i];
    G_HIP(hipSetDevice(m.device));
    if (g->coll == &kPeerCopy)   // the send block is still being pulled by the entry's previous batch until every member says otherwise
      for (Member& o : g->members)
        if (o.pulled_recorded[(size_t)k]) G_HIP(hipStreamWaitEvent(m.comm_stream, o.pulled[(size_t)k], 0));
    if (!e.enqueued[(size_t)i]) {   // no frames of this batch fell to the member's frame group: an empty block
      G_HIP(hipMemsetAsync(m.d_send[(size_t)k], 0, LMX_GATHER_HEADER_BYTES, m.comm_stream));
      return LMX_OK;
    }
    return oldest ? lmx_ctx_export_oldest_on(m.ctx, m.d_send[(size_t)k], e.capacity, m.comm_stream)
                  : lmx_ctx_export_raw_on(m.ctx, m.d_send[(size_t)k], e.capacity, m.comm_stream);
    });
  }
  if (st != LMX_OK) return st;
  {
    Phase ph(g, lmx_group::T_GATHER);
    if ((st = g->coll->all_gather(g, k)) != LMX_OK) return st;
  }
  Phase ph(g, lmx_group::T_COPY);
  Member& m0 = g->members[0];
  G_HIP(hipSetDevice(m0.device));
  st = lmx_stream_copy_blocks(e.h_blocks, m0.d_recv[(size_t)k], g->world, lmx_group::block_bytes(e.capacity), e.capacity, m0.comm_stream);
  if (st != LMX_OK) return st;
  G_HIP(hipEventRecord(e.ready, m0.comm_stream));
  return LMX_OK;
}

}  // namespace

extern "C" {

lmx_status lmx_group_unique_id(void* out128) {
  if (!out128) { lmx::set_error("lmx_group_unique_id: null argument"); return LMX_ERR_INVALID_ARG; }
  Rccl* R = rccl();
  if (!R) { lmx::set_error("librccl.so could not be loaded: %s", g_rccl.load_error.c_str()); return LMX_ERR_NOT_FOUND; }
  ncclUniqueId id;
  G_NCCL(R->GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
  return LMX_OK;
}

void lmx_group_destroy(lmx_group* g) {
  if (!g) return;
  if (g->trace) {
    static const char* names[lmx_group::T_COUNT] = {"upload:begin", "upload:stage", "upload:dma", "submit:enqueue", "submit:export", "submit:all_gather", "submit:copy",
                                                    "finish:wait", "finish:merge", "finish:release"};
    std::fprintf(stderr, "lmx_group trace (%d members, %s): host us per call:", (int)g->members.size(), g->coll ? g->coll->name : "?");
    for (int i = 0; i < lmx_group::T_COUNT; ++i)
      if (g->t_n[i]) std::fprintf(stderr, " %s %.1f", names[i], g->t_acc[i] / (double)g->t_n[i] * 1e6);
    std::fprintf(stderr, "\n");
  }
  for (Member& m : g->members) {
    (void)hipSetDevice(m.device);
    if (m.comm_stream) (void)hipStreamSynchronize(m.comm_stream);
  }
  if (g->coll) g->coll->destroy(g);
  for (int k = 0; k < (int)g->ring.size(); ++k) {
    free_entry(g, k);
    if (g->ring[(size_t)k].ready) (void)hipEventDestroy(g->ring[(size_t)k].ready);
  }
  for (Member& m : g->members) {
    (void)hipSetDevice(m.device);
    if (m.ctx) lmx_ctx_destroy(m.ctx);
    for (hipEvent_t e : m.sent) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m.pulled) if (e) (void)hipEventDestroy(e);
    if (m.comm_stream) (void)hipStreamDestroy(m.comm_stream);
  }
  for (uint8_t* p : g->h_stage) if (p) (void)hipHostFree(p);
  delete g;
}

lmx_status lmx_group_create(const lmx_bank* bank, const lmx_group_desc* desc, lmx_group** out) {
  return lmx::guarded("lmx_group_create", [&]() -> lmx_status {
  if (!bank || !desc || !out) { lmx::set_error("lmx_group_create: null argument"); return LMX_ERR_INVALID_ARG; }
  const bool multi_process = desc->unique_id != nullptr;
  if (multi_process && (desc->world < 1 || desc->rank < 0 || desc->rank >= desc->world)) { lmx::set_error("lmx_group_create: rank %d outside world %d", desc->rank, desc->world); return LMX_ERR_INVALID_ARG; }
  if (!multi_process && desc->n_devices < 1) { lmx::set_error("lmx_group_create: n_devices must be >= 1"); return LMX_ERR_INVALID_ARG; }
  if (desc->max_batch < 1) { lmx::set_error("lmx_group_create: max_batch must be >= 1"); return LMX_ERR_INVALID_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { lmx::set_error("no HIP device available; this library has no CPU path"); return LMX_ERR_NO_DEVICE; }
  int collective = desc->collective;
  if (const char* e = std::getenv("LMX_GROUP_COLLECTIVE")) {   // read once, here
    if (std::strcmp(e, "peer") == 0 || std::strcmp(e, "peer_copy") == 0) collective = LMX_GROUP_COLLECTIVE_PEER_COPY;
    else if (std::strcmp(e, "rccl") == 0) collective = LMX_GROUP_COLLECTIVE_RCCL;
  }
  const bool trace = std::getenv("LMX_GROUP_TRACE") != nullptr;
  if (collective != LMX_GROUP_COLLECTIVE_RCCL && collective != LMX_GROUP_COLLECTIVE_PEER_COPY) { lmx::set_error("lmx_group_create: unknown collective %d", collective); return LMX_ERR_INVALID_ARG; }
  lmx_group* g = new lmx_group();
  g->bank = bank; g->desc = *desc; g->multi_process = multi_process; g->trace = trace;
  g->coll = collective == LMX_GROUP_COLLECTIVE_PEER_COPY ? &kPeerCopy : &kRccl;
  g->world = multi_process ? desc->world : desc->n_devices;
  g->G = desc->frame_groups > 1 ? desc->frame_groups : 1;
  if (g->world % g->G != 0) { lmx::set_error("lmx_group_create: %d members do not form %d frame groups (frame_groups must divide the member count)", g->world, g->G); delete g; return LMX_ERR_INVALID_ARG; }
  g->R = g->world / g->G;
  const int n_local = multi_process ? 1 : desc->n_devices;
  g->members.resize((size_t)n_local);
  g->devlist.resize((size_t)n_local);
  for (int i = 0; i < n_local; ++i) {
    Member& m = g->members[(size_t)i];
    m.device = multi_process ? desc->device : (desc->devices ? desc->devices[i] : i);
    m.rank = multi_process ? desc->rank : i;
    m.fgroup = m.rank / g->R; m.shard = m.rank % g->R;
    g->devlist[(size_t)i] = m.device;
    if (m.device < 0 || m.device >= ndev) { lmx::set_error("device %d out of range (%d devices)", m.device, ndev); lmx_group_destroy(g); return LMX_ERR_NO_DEVICE; }
  }
  auto fail = [&](lmx_status st) { std::string keep = lmx_last_error(); lmx_group_destroy(g); lmx::set_error("%s", keep.c_str()); return st; };
  // The collective first, and its first exchange right away: RCCL sets up its streams, channels and buffers lazily at a communicator's
  // first collective, and when that happens after the members' contexts (their lane streams) exist the matching kernels run 2.5 % slower
  // from then on (measured with one rank, 138.5 k against 142.1 k frames/s: scripts/sharded_overhead_split.py).
  {
    lmx_status st0 = g->coll->init(g);
    if (st0 != LMX_OK) return fail(st0);
    if (g->coll->warm_up && (st0 = g->coll->warm_up(g)) != LMX_OK) return fail(st0);
  }
  for (Member& m : g->members) {
    if (hipSetDevice(m.device) != hipSuccess) return fail(LMX_ERR_HIP);
    lmx_ctx_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.device = m.device; cd.width = desc->width; cd.height = desc->height; cd.max_candidates = desc->max_candidates;
    cd.max_batch = (desc->max_batch + g->G - 1) / g->G;   // a member only ever holds its frame group's share of a batch
    cd.shard_rank = m.shard; cd.shard_world = g->R;
    cd.flags = desc->flags | (n_local > 1 ? lmx::LMX_CTX_EXTERNAL_STAGING : 0);   // several members: the group stages once for all of them
    lmx_status st = lmx_ctx_create(bank, &cd, &m.ctx);
    if (st != LMX_OK) return fail(st);
    if (hipStreamCreateWithFlags(&m.comm_stream, hipStreamNonBlocking) != hipSuccess) { lmx::set_error("hipStreamCreate failed"); return fail(LMX_ERR_HIP); }
  }
  g->depth = lmx_ctx_max_outstanding(g->members[0].ctx);
  // every member's per-entry vectors exist before the ring does: lmx_group_destroy (the failure path below included) walks ring x members
  for (Member& m : g->members) {
    m.d_send.assign((size_t)g->depth, nullptr); m.d_recv.assign((size_t)g->depth, nullptr);
    m.sent.assign((size_t)g->depth, nullptr); m.pulled.assign((size_t)g->depth, nullptr); m.pulled_recorded.assign((size_t)g->depth, 0);
  }
  g->ring.resize((size_t)g->depth);
  for (RingEntry& e : g->ring) e.enqueued.assign(g->members.size(), 0);
  for (Member& m : g->members) {
    if (hipSetDevice(m.device) != hipSuccess) return fail(LMX_ERR_HIP);
    for (int k = 0; k < g->depth; ++k)
      if (hipEventCreateWithFlags(&m.sent[(size_t)k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m.pulled[(size_t)k], hipEventDisableTiming) != hipSuccess) {
        lmx::set_error("hipEventCreate failed");
        return fail(LMX_ERR_HIP);
      }
  }
  if (hipSetDevice(g->members[0].device) != hipSuccess) return fail(LMX_ERR_HIP);
  for (int k = 0; k < g->depth; ++k)
    if (hipEventCreateWithFlags(&g->ring[(size_t)k].ready, hipEventDisableTiming) != hipSuccess) { lmx::set_error("hipEventCreate failed"); return fail(LMX_ERR_HIP); }
  lmx_status st = LMX_OK;
  g->capacity = desc->gather_capacity > 0 ? (size_t)desc->gather_capacity : 8192;
  for (int k = 0; k < g->depth; ++k)
    if ((st = alloc_entry(g, k, g->capacity)) != LMX_OK) return fail(st);
  if (n_local > 1) {
    // host threads: one per member (at most 8) drive the members' uploads / enqueues / exports in parallel -- issued from one
    // thread, eight members cost ~0.4 ms of launch calls per batch, as long as a 64-frame batch runs on the device -- and the
    // same pool does the one staging copy of a batch of host frames (LMX_GROUP_THREADS overrides, read once here)
    const unsigned hw = std::thread::hardware_concurrency();
    int threads = std::max(std::min(n_local, 8), (int)std::max(1u, std::min(8u, hw ? hw / 2 : 1u)));   // the staging copy wants a few threads whatever the member count
    if (const char* e = std::getenv("LMX_GROUP_THREADS")) { const int n = std::atoi(e); if (n >= 1) threads = std::min(n, 64); }
    g->pool.reset(new lmx::CopyPool(threads - 1));
    g->h_stage.assign((size_t)lmx::ctx_num_sets(g->members[0].ctx), nullptr);
  }
  *out = g;
  return LMX_OK;
  });
}

int32_t lmx_group_size(const lmx_group* g) { return g ? g->world : 0; }
int32_t lmx_group_frame_groups(const lmx_group* g) { return g ? g->G : 0; }
int32_t lmx_group_gather_capacity(const lmx_group* g) { return g ? (int32_t)g->capacity : 0; }
int32_t lmx_group_depth(const lmx_group* g) { return g ? g->depth : 0; }
const char* lmx_group_collective_name(const lmx_group* g) { return g && g->coll ? g->coll->name : ""; }

lmx_status lmx_group_upload(lmx_group* g, int32_t n_frames, const lmx_image* sources, int32_t n_sources) {
  return lmx::guarded("lmx_group_upload", [&]() -> lmx_status {
  if (!g || !sources) { lmx::set_error("lmx_group_upload: null argument"); return LMX_ERR_INVALID_ARG; }
  if (n_frames < 1 || n_frames > g->desc.max_batch) { lmx::set_error("n_frames=%d outside [1,%d]", n_frames, g->desc.max_batch); return LMX_ERR_INVALID_ARG; }
  if (g->members.size() == 1) {   // one member per process: the context's own upload path (staging, or direct stores for one or two frames)
    Member& m = g->members[0];
    int first = 0, count = 0;
    g->frames_of(m.fgroup, n_frames, &first, &count);
    if (n_sources < 1) { lmx::set_error("lmx_group_upload: no sources"); return LMX_ERR_SHAPE; }
    // its frame group's slice of the batch (all of it when G == 1); a batch with fewer frames than groups may leave this member idle
    lmx_status st = count > 0 ? lmx_ctx_upload(m.ctx, count, sources + (size_t)first * (size_t)n_sources, n_sources) : LMX_OK;
    if (st == LMX_OK) { g->uploaded = true; g->n_uploaded = n_frames; }
    return st;
  }
  lmx_ctx* c0 = g->members[0].ctx;
  lmx_status st = lmx::ctx_check_sources(c0, n_frames, sources, n_sources, g->desc.max_batch);
  if (st != LMX_OK) return st;
  const int set = lmx::ctx_next_set(c0);
  for (Member& m : g->members)
    if (lmx::ctx_next_set(m.ctx) != set) { lmx::set_error("lmx_group_upload: the members' frame sets are out of step (a member context was used outside the group)"); return LMX_ERR_INVALID_ARG; }
  // every member: the previous transfer out of staging area `set` has finished (host), its lanes are done with frame set `set` (device)
  {
    Phase ph(g, lmx_group::T_BEGIN);
    st = for_members(g, [&](int i) { return lmx::ctx_begin_staged_upload(g->members[(size_t)i].ctx); });
  }
  if (st != LMX_OK) return st;
  if (!g->h_stage[(size_t)set]) {
    G_HIP(hipSetDevice(g->members[0].device));
    G_HIP(hipHostMalloc((void**)&g->h_stage[(size_t)set], lmx::ctx_bytes_per_frame(c0) * (size_t)g->desc.max_batch, hipHostMallocPortable));   // every device DMAs from it
  }
  // ONE staging copy of the whole batch (non-temporal stores, the group's host threads), then every member transfers its frame group's
  // slice of it over its own link (G == 1: N transfers of the same bytes)
  {
    Phase ph(g, lmx_group::T_STAGE);
    lmx::ctx_stage_sources(c0, g->pool.get(), g->h_stage[(size_t)set], n_frames, sources, g->desc.max_batch);
  }
  Phase ph(g, lmx_group::T_DMA);
  st = for_members(g, [&](int i) {
    Member& m = g->members[(size_t)i];
    int first = 0, count = 0;
    g->frames_of(m.fgroup, n_frames, &first, &count);
    return lmx::ctx_finish_staged_upload(m.ctx, count, g->h_stage[(size_t)set], first, g->desc.max_batch);
  });
  if (st == LMX_OK) { g->uploaded = true; g->n_uploaded = n_frames; }
  return st;
  });
}

lmx_status lmx_group_submit(lmx_group* g, int32_t n_frames, float threshold, const char* const* class_ids, int32_t n_class_ids) {
  return lmx::guarded("lmx_group_submit", [&]() -> lmx_status {
  if (!g) { lmx::set_error("lmx_group_submit: null group"); return LMX_ERR_INVALID_ARG; }
  if (!g->uploaded) { lmx::set_error("lmx_group_submit: nothing uploaded"); return LMX_ERR_INVALID_ARG; }
  if (g->in_flight >= g->depth) { lmx::set_error("lmx_group_submit: %d batches are already in flight; finish one first", g->in_flight); return LMX_ERR_INVALID_ARG; }
  const int k = g->head;
  RingEntry& e = g->ring[(size_t)k];
  lmx_status st = LMX_OK;
  if (e.capacity < g->capacity && (st = alloc_entry(g, k, g->capacity)) != LMX_OK) return st;   // an earlier batch made the blocks grow
  if (n_frames < 1 || n_frames > g->n_uploaded) { lmx::set_error("lmx_group_submit: n_frames=%d but the most recent upload holds %d frame(s)", n_frames, g->n_uploaded); return LMX_ERR_INVALID_ARG; }
  // a member's share of the batch is its frame group's slice of the UPLOADED frames, cut off at n_frames (frames are dealt at upload time)
  auto share = [&](const Member& m) {
    int first = 0, count = 0;
    g->frames_of(m.fgroup, g->n_uploaded, &first, &count);
    return std::max(0, std::min(count, n_frames - first));
  };
  if (g->G > 1 && n_frames != g->n_uploaded) { lmx::set_error("lmx_group_submit: a group with frame groups matches the whole upload (%d frames), not %d", g->n_uploaded, n_frames); return LMX_ERR_INVALID_ARG; }
  std::vector<char> enqueued(g->members.size(), 0);
  // graph captures (first use of a slot / frame set / batch size / threshold) run here, one after the other on the calling thread: a
  // capture that overlaps HIP calls of the group's other host threads fails on ROCm 7.2 (profiles/r03_group_host_cost.txt)
  for (Member& m : g->members)
    if (share(m) > 0 && (st = lmx::ctx_prepare_graph(m.ctx, share(m), threshold)) != LMX_OK) return st;
  {
    Phase ph(g, lmx_group::T_ENQUEUE);
    st = for_members(g, [&](int i) {
      const int n_mine = share(g->members[(size_t)i]);
      if (n_mine == 0) return LMX_OK;
      lmx_status r = lmx_ctx_enqueue(g->members[(size_t)i].ctx, n_mine, threshold, class_ids, n_class_ids);
      enqueued[(size_t)i] = r == LMX_OK;
      return r;
    });
  }
  e.enqueued = enqueued;
  if (st == LMX_OK) st = queue_exchange(g, k, false);
  if (st != LMX_OK) {
    // take back what was queued on the members that did enqueue: their output slots must not stay outstanding
    const std::string keep = lmx_last_error();
    for (size_t i = 0; i < g->members.size(); ++i)
      if (enqueued[i]) (void)lmx::ctx_drop_newest(g->members[i].ctx);
    e.enqueued.assign(g->members.size(), 0);
    lmx::set_error("%s", keep.c_str());
    return st;
  }
  e.n_frames = n_frames;
  g->head = (k + 1) % g->depth;
  g->in_flight += 1;
  return LMX_OK;
  });
}

lmx_status lmx_group_finish(lmx_group* g, int32_t n_frames, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_group_finish", [&]() -> lmx_status {
  if (!g || !n_out || (cap > 0 && !out)) { lmx::set_error("lmx_group_finish: null argument"); return LMX_ERR_INVALID_ARG; }
  if (g->in_flight < 1) { lmx::set_error("lmx_group_finish: nothing submitted"); return LMX_ERR_INVALID_ARG; }
  const int k = (g->head + g->depth - g->in_flight) % g->depth;   // oldest batch in flight
  RingEntry& e = g->ring[(size_t)k];
  if (n_frames != e.n_frames) { lmx::set_error("lmx_group_finish: n_frames=%d but the batch was submitted with %d", n_frames, e.n_frames); return LMX_ERR_INVALID_ARG; }
  auto finish_members = [&](lmx_status st) {   // frees the members' output slots whether or not the exchange worked
    const std::string keep = st != LMX_OK ? lmx_last_error() : "";
    for (size_t i = 0; i < g->members.size(); ++i) {
      if (!e.enqueued[i]) continue;   // the member had no frames in this batch: nothing of it is outstanding
      const lmx_status rs = lmx_ctx_release(g->members[i].ctx);
      if (st == LMX_OK && rs != LMX_OK) st = rs;
    }
    if (!keep.empty()) lmx::set_error("%s", keep.c_str());
    g->in_flight -= 1;
    return st;
  };
  lmx_status st = LMX_OK;
  {
    Phase ph(g, lmx_group::T_WAIT);
    hipError_t he = hipSetDevice(g->members[0].device);
    if (he == hipSuccess) he = hipEventSynchronize(e.ready);
    if (he != hipSuccess) { lmx::set_error("waiting for the exchange failed: %s", hipGetErrorString(he)); st = LMX_ERR_HIP; }
  }
  // two-phase fallback: the headers say how many records every rank really has
  if (st == LMX_OK) {
    size_t need = 0;
    const size_t bb = lmx_group::block_bytes(e.capacity);
    for (int r = 0; r < g->world; ++r) need = std::max<size_t>(need, reinterpret_cast<const uint32_t*>(e.h_blocks + (size_t)r * bb)[1]);
    if (need > e.capacity) {
      size_t grown = std::max<size_t>(e.capacity, 1);
      while (grown < need) grown *= 2;
      g->capacity = std::max(g->capacity, grown);   // later batches allocate at this size when their ring entry comes round
      st = alloc_entry(g, k, grown);
      if (st == LMX_OK) st = queue_exchange(g, k, true);   // this batch is the oldest outstanding enqueue of every member
      if (st == LMX_OK) {
        hipError_t he = hipEventSynchronize(e.ready);
        if (he != hipSuccess) { lmx::set_error("waiting for the repeated exchange failed: %s", hipGetErrorString(he)); st = LMX_ERR_HIP; }
      }
    }
  }
  std::vector<size_t> offsets((size_t)n_frames + 1, 0);
  std::vector<lmx_match_t> flat;
  if (st == LMX_OK) {
    Phase ph(g, lmx_group::T_MERGE);
    const size_t bb = lmx_group::block_bytes(e.capacity);
    flat.resize(std::max<size_t>(1, cap * (size_t)n_frames));
    st = lmx_merge_gathered_groups(e.h_blocks, g->world, bb, e.capacity, n_frames, g->G, flat.data(), flat.size(), offsets.data());
    if (st == LMX_ERR_OVERFLOW && offsets[(size_t)n_frames] > flat.size()) {   // more matches than cap * n_frames in total: size exactly, report per frame below
      flat.resize(offsets[(size_t)n_frames]);
      st = lmx_merge_gathered_groups(e.h_blocks, g->world, bb, e.capacity, n_frames, g->G, flat.data(), flat.size(), offsets.data());
    }
  }
  {
    Phase ph(g, lmx_group::T_RELEASE);
    st = finish_members(st);
  }
  if (st != LMX_OK) { for (int f = 0; f < n_frames; ++f) n_out[f] = 0; return st; }
  for (int f = 0; f < n_frames; ++f) {
    const size_t n = offsets[(size_t)f + 1] - offsets[(size_t)f];
    n_out[f] = n;
    if (cap) std::memcpy(out + (size_t)f * cap, flat.data() + offsets[(size_t)f], std::min(n, cap) * sizeof(lmx_match_t));
    if (n > cap) { lmx::set_error("frame %d: %zu matches > output capacity %zu", f, n, cap); st = LMX_ERR_OVERFLOW; }
  }
  return st;
  });
}

lmx_status lmx_group_match_batch(lmx_group* g, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                                 const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_group_match_batch", [&]() -> lmx_status {
  if (!g || !sources || !n_out || (cap > 0 && !out)) { lmx::set_error("lmx_group_match_batch: null argument"); return LMX_ERR_INVALID_ARG; }
  if (g->in_flight != 0) { lmx::set_error("lmx_group_match_batch: %d submitted batches are still in flight; finish them first", g->in_flight); return LMX_ERR_INVALID_ARG; }
  // G == 1: every rank sees the same frames (pre-processing is replicated: cheaper than moving linear memories over xGMI); G > 1: the
  // frames are dealt to the frame groups
  lmx_status st = lmx_group_upload(g, n_frames, sources, n_sources);
  if (st == LMX_OK) st = lmx_group_submit(g, n_frames, threshold, class_ids, n_class_ids);
  if (st != LMX_OK) return st;
  return lmx_group_finish(g, n_frames, out, cap, n_out);
  });
}

}  // extern "C"
