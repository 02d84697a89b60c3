// dist.hip -- multi-GPU transport (RCCL over xGMI, or host callbacks) and the domain-decomposed BBPGD solve driven
// from C++.  Host code only: the kernels it sequences live in convex.hip / reorder.hip and are reached
// through the public stage entry points, so this loop and a host that drives the stages by hand produce the same
// iterates.
//
// Reference counterparts: the MPI communicator of stk::search::coarse_search / change_ghosting
// (mundy_mesh/GenNeighborLinkers.hpp:658, :687-711); per iteration stk::all_reduce_max + 3 x stk::all_reduce_sum
// (scrap/lcp_spheres/NGPSpheresLCP.cpp:371, :450-452) and the ghost refresh left as a TODO at :1057.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is looked up at run time

#include <cctype>
#include <initializer_list>
#include <string>
#include <vector>

#include "mhip_internal.hpp"

namespace mhip {

struct RcclApi {
  void* lib = nullptr;
  bool tried = false;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok() const {
    return GetUniqueId && CommInitRank && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd &&
           GetErrorString;
  }
};

static RcclApi& rccl() {
  static RcclApi api;
  if (api.tried) return api;
  api.tried = true;
  // a process that already carries an RCCL (PyTorch ships its own librccl.so.1) must keep using that one
  api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
    if (api.lib) break;
    api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!api.lib) return api;
#define MHIP_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
  MHIP_SYM(GetUniqueId, "ncclGetUniqueId");
  MHIP_SYM(CommInitRank, "ncclCommInitRank");
  MHIP_SYM(CommDestroy, "ncclCommDestroy");
  MHIP_SYM(AllGather, "ncclAllGather");
  MHIP_SYM(Send, "ncclSend");
  MHIP_SYM(Recv, "ncclRecv");
  MHIP_SYM(GroupStart, "ncclGroupStart");
  MHIP_SYM(GroupEnd, "ncclGroupEnd");
  MHIP_SYM(GetErrorString, "ncclGetErrorString");
#undef MHIP_SYM
  return api;
}

#define MHIP_RCCL(call)                                                                                         \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess)                                                                                      \
      return ::mhip::fail(MHIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, \
                          __LINE__);                                                                            \
  } while (0)

}  // namespace mhip

using namespace mhip;

struct mhip_comm {
  int rank = 0, world = 1;
  bool is_rccl = false;
  // RCCL transport: the grouped send / recv runs on comm_stream (so it overlaps the caller's kernels); `ready`
  // carries the caller's stream into it, `done` carries it back.  Collectives run on the caller's stream.
  ncclComm_t nccl = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  bool in_flight = false;
  // host-callback transport: the exchange is kept until finish (the callbacks block)
  mhip_comm_exchange_fn xfn = nullptr;
  mhip_comm_all_gather_fn gfn = nullptr;
  void* user = nullptr;
  std::vector<int> send_peer, recv_peer;
  std::vector<const double*> send_buf;
  std::vector<double*> recv_buf;
  std::vector<size_t> send_count, recv_count;
  // Mailbox (mhip_comm_mailbox_open): the per-iteration record exchange of the ranks of ONE node through slots in each
  // other's device memory -- every rank owns a box of fine-grained memory, IPC-mapped by all the others; a rank writes
  // its record into its slot of every box and polls its own.  No collective launch on the critical path of an iteration.
  struct Mailbox {
    bool open = false;
    void* own = nullptr;             // this rank's box
    std::vector<void*> mapped;       // the other ranks' boxes as mapped here (null at this rank's place)
    DeviceBuffer peers;              // [world] device array of box pointers
    DeviceBuffer status;             // [0] != 0: an exchange timed out (sticky); [1] = exchanges made (device-counted)
  } mbox;
  // Velocity halo without the collective library (mhip_comm_halo_ipc_enable): every rank owns an INBOX of fine-grained
  // memory, one slot of kHaloWords 8-byte words per local body row, IPC-mapped by all the others.  After its body sweep
  // a rank writes the rows its peers hold as ghosts straight into their inboxes (posted writes over xGMI: one kernel, no
  // send / recv launch, no stream hand-over); before its boundary sweep a rank collects its ghost rows from its own
  // inbox (local polling reads) into the velocity table.  Every word carries 32 bits of data and the number of the
  // exchange, so each store validates itself and the writes may land in any order (the mailbox's protocol).
  struct HaloIpc {
    bool wanted = false, open = false, plan_ok = false;
    void* own = nullptr;             // this rank's inbox
    size_t capacity = 0;             // rows it holds
    std::vector<void*> mapped;       // the other ranks' inboxes as mapped here
    std::vector<unsigned long long*> base;  // [world] inbox pointers as seen from this rank
    uint32_t seq_base = 1;           // number of the first exchange of the next solve (the same on every rank:
                                     // advanced at ONE place per solve whatever its outcome, re-agreed -- the
                                     // maximum over the ranks -- by every ghost plan; never 0, the cleared inbox's tag)
    // of the current ghost plan: per send peer the first row of my block in ITS velocity table
    std::vector<size_t> dst_first_row;
  } hipc;
  // bound of every wait on a peer's words (mailbox, inboxes), in ticks of the 100 MHz wall clock
  unsigned long long timeout_ticks = 2000000000ull;
  // TEST HOOK (mhip_comm_inject_fault): the next distributed solve fails at this convergence poll (0 = off)
  unsigned fault_at_poll = 0;
  // what the last distributed solve used (mhip_dist_profile reports it)
  int last_halo_path = 0, last_record_path = 0;
  // work buffers of the distributed solve
  DeviceBuffer send_rows, triples;
  std::vector<hipEvent_t> events;
  // ghost plan of the last mhip_ghost_plan: who gets which owned rows, where the ghosts' rows land
  struct GhostPlan {
    bool valid = false;
    size_t n = 0, n_lo = 0, n_hi = 0, total_send = 0;
    std::vector<int> send_peer, recv_peer;
    std::vector<size_t> send_rows, recv_first_row, recv_rows;
    DeviceBuffer send_index;        // owned indices (0-based in the owned block), peer after peer
    DeviceBuffer send_index_local;  // the same as local indices (+ n_lo): what the velocity halo gathers
    DeviceBuffer stage;             // counts on their way through the all-gather; packed send rows
    DeviceBuffer regions;           // this rank's chunk boxes + everybody's, all-gathered
  } ghost;
  // migration plan of the last mhip_migrate_plan: which owned rows leave for which rank, how many arrive from whom
  struct MigratePlan {
    bool valid = false;
    size_t n = 0, n_new = 0, keep = 0, keep_first = 0;
    std::vector<int> send_peer, recv_peer;
    std::vector<size_t> send_first_row, send_rows, recv_rows;
    DeviceBuffer dest, sortkey, order, hist, stage;
  } migrate;
};

namespace mhip {
__global__ void __launch_bounds__(kBlock) k_offset_i32(size_t n, const int32_t* __restrict__ in, int32_t off,
                                                      int32_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) out[i] = in[i] + off;
}
// hist[key[i]] += weight[i] (1 if weights is null).  Weights are body counts / contact counts: integer-valued, so the
// sums are exact whatever order the atomics land in.
__global__ void __launch_bounds__(kBlock) k_weighted_hist(size_t n, const uint32_t* __restrict__ keys,
                                                         const double* __restrict__ weights, size_t nbins,
                                                         double* __restrict__ hist) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const size_t k = keys[i] < nbins ? keys[i] : nbins - 1;
    atomicAdd(&hist[k], weights ? weights[i] : 1.0);
  }
}
// dest[i] = number of splitters below key[i] (rank r owns the cells  splitters[r - 1] < key <= splitters[r]);
// sortkey = dest << 32 | i, so a stable sort groups the rows by destination in their present order
__global__ void __launch_bounds__(kBlock) k_migrate_dest(size_t n, const uint32_t* __restrict__ keys, int nsplit,
                                                        const long long* __restrict__ splitters,
                                                        unsigned long long* __restrict__ sortkey,
                                                        double* __restrict__ count /*[nsplit + 1]*/) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const long long key = keys[i];
    int lo = 0, hi = nsplit;  // first index with splitters[idx] >= key
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (splitters[mid] < key) lo = mid + 1; else hi = mid;
    }
    sortkey[i] = (static_cast<unsigned long long>(lo) << 32) | static_cast<unsigned long long>(i);
    atomicAdd(&count[lo], 1.0);
  }
}
// all-gather of a few host doubles through the communicator: out [world][count]
static int host_all_gather(mhip_comm* c, const double* in, size_t count, std::vector<double>& out, hipStream_t s) {
  const size_t w = (size_t)c->world;
  if (int e = c->ghost.stage.reserve((count + w * count + 2) * sizeof(double))) return e;
  double* d_in = c->ghost.stage.as<double>();
  double* d_out = d_in + count;
  MHIP_HIP(hipMemcpyAsync(d_in, in, count * sizeof(double), hipMemcpyHostToDevice, s));
  if (int e = mhip_comm_all_gather(c, d_in, count, d_out, reinterpret_cast<mhip_stream_t>(s))) return e;
  out.resize(w * count);
  MHIP_HIP(hipMemcpyAsync(out.data(), d_out, w * count * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  return MHIP_SUCCESS;
}
// the exchange as a launch of its own (the trial exchanges of mhip_comm_mailbox_open; the solve has it inside the
// kernel that forms the record, mhip_bbpgd_stage_reduce_exchange)
__global__ void __launch_bounds__(64) k_mailbox_exchange(MailboxArgs m, const double* __restrict__ local) {
  mailbox_exchange_wave(m, local);
}
// (default of mhip_comm::timeout_ticks: 20 s of the 100 MHz wall clock -- ranks enter a solve at different times: the
//  narrow phase of a mixed system is uneven; mhip_comm_set_exchange_timeout changes it)

// ---- velocity halo through IPC-mapped inboxes -----------------------------------------------------------------------
constexpr int kHaloWords = 12;      // a row of 6 doubles as 12 (data, exchange number) words
constexpr int kHaloMaxPeers = 16;   // ranks of one node
struct HaloPushArgs {
  int npeers = 0;
  unsigned long long* base[kHaloMaxPeers];  // peer k's inbox
  unsigned first[kHaloMaxPeers + 1];        // send rows [first[k], first[k + 1]) go to peer k ...
  unsigned dst_row[kHaloMaxPeers];          // ... into its rows dst_row[k] + (r - first[k])
};
// exchange number of this launch: seq_base + (init ? 0 : flips + 1); flips only changes in the finalize launches
__device__ inline unsigned halo_seq(unsigned seq_base, int init, const unsigned* flips) {
  return seq_base + (init ? 0u : *flips + 1u);
}
// one lane per (send row, word): consecutive lanes write consecutive words of a peer's inbox
__global__ void __launch_bounds__(kBlock)
    k_halo_push(HaloPushArgs a, size_t total_rows, const int32_t* __restrict__ send_index,
                const double* __restrict__ vel, unsigned seq_base, int init, const unsigned* __restrict__ flips,
                const int* __restrict__ done) {
  if (!init && *done) return;
  const unsigned seq = halo_seq(seq_base, init, flips);
  const size_t nw = total_rows * kHaloWords;
  for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < nw; t += (size_t)gridDim.x * kBlock) {
    const unsigned r = static_cast<unsigned>(t / kHaloWords);
    const int w = static_cast<int>(t % kHaloWords);
    int k = 0;
    while (k + 1 < a.npeers && r >= a.first[k + 1]) ++k;
    const unsigned long long bits =
        static_cast<unsigned long long>(__double_as_longlong(vel[6 * static_cast<size_t>(send_index[r]) + (w >> 1)]));
    const unsigned half = static_cast<unsigned>((w & 1) ? (bits >> 32) : (bits & 0xffffffffull));
    unsigned long long* dst = a.base[k] + (static_cast<size_t>(a.dst_row[k]) + (r - a.first[k])) * kHaloWords + w;
    __hip_atomic_store(dst, (static_cast<unsigned long long>(seq) << 32) | half, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// one lane per (ghost row, word): polls its word until it carries this exchange's number (bounded), the even lane of a
// pair writes the double into the velocity table.  Ghost rows = the local rows outside the owned block.
__global__ void __launch_bounds__(kBlock)
    k_halo_collect(const unsigned long long* __restrict__ inbox, size_t n_lo, size_t n_owned, size_t n_ghost,
                   double* __restrict__ vel, unsigned seq_base, int init, const unsigned* __restrict__ flips,
                   const int* __restrict__ done, unsigned long long* __restrict__ status,
                   unsigned long long timeout) {
  if (!init && *done) return;
  const unsigned seq = halo_seq(seq_base, init, flips);
  const size_t nw = n_ghost * kHaloWords;
  for (size_t t0 = blockIdx.x * (size_t)kBlock; t0 < nw; t0 += (size_t)gridDim.x * kBlock) {
    const size_t t = t0 + threadIdx.x;   // (t0 and kHaloWords are even: lanes (2m, 2m + 1) of a wave hold the two halves
    const bool live = t < nw;            //  of one double, and both are live or both are not)
    unsigned long long word = 0;
    bool ok = true, overrun = false;
    size_t row = 0;
    int w = 0;
    if (live) {
      const size_t o = t / kHaloWords;
      w = static_cast<int>(t % kHaloWords);
      row = (o < n_lo) ? o : o + n_owned;
      const unsigned long long* src = inbox + row * kHaloWords + w;
      const unsigned long long start = wall_clock64();
      for (;;) {
        word = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned tag = static_cast<unsigned>(word >> 32);
        if (tag == seq) break;
        // One slot per ghost row: a peer's push of exchange k + 1 must not land before this rank has collected exchange
        // k.  The iteration's own dependencies give that order (a peer pushes k + 1 after ITS finalize of k, which
        // needs this rank's record of k, formed after this collect): a word that is AHEAD of the exchange being
        // collected means the order was broken -- reported, never consumed.
        if (tag != 0u && tag - seq - 1u < 0x3fffffffu) {
          overrun = true;
          ok = false;
          break;
        }
        if (__hip_atomic_load(&status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull ||
            wall_clock64() - start > timeout) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) __hip_atomic_store(&status[0], overrun ? 2ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // t even <-> w even (kHaloWords is even): lane pairs (2m, 2m + 1) hold the two halves of one double
    const unsigned long long other = __shfl_xor(word, 1, 64);
    const int ok_other = __shfl_xor(ok ? 1 : 0, 1, 64);
    if (live && !(w & 1)) {
      const unsigned long long bits = (other << 32) | (word & 0xffffffffull);
      vel[6 * row + (w >> 1)] = (ok && ok_other) ? __longlong_as_double(static_cast<long long>(bits)) : __builtin_nan("");
    }
  }
}
void halo_ipc_close(mhip_comm* c) {
  auto& h = c->hipc;
  for (void* p : h.mapped)
    if (p) (void)hipIpcCloseMemHandle(p);
  if (h.own) (void)hipFree(h.own);
  h.mapped.clear();
  h.base.clear();
  h.own = nullptr;
  h.capacity = 0;
  h.open = false;
  h.plan_ok = false;
}
static int host_all_gather(mhip_comm* c, const double* in, size_t count, std::vector<double>& out, hipStream_t s);
// (Re)opens the inboxes with room for `rows` rows on every rank.  Collective; every rank takes the same decisions.
int halo_ipc_open(mhip_comm* c, size_t rows, hipStream_t s) {
  auto& h = c->hipc;
  halo_ipc_close(c);
  const int world = c->world;
  const size_t bytes = (rows * kHaloWords + (size_t)world) * sizeof(unsigned long long);  // + one probe word per rank
  constexpr size_t kHandleDoubles = (sizeof(hipIpcMemHandle_t) + 7) / 8;
  double mine_ok = 0.0;
  hipIpcMemHandle_t handle;
  memset(&handle, 0, sizeof(handle));
  void* own = nullptr;
  if (hipExtMallocWithFlags(&own, bytes, hipDeviceMallocFinegrained) == hipSuccess && own != nullptr) {
    if (hipMemsetAsync(own, 0, bytes, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess &&
        (world == 1 || hipIpcGetMemHandle(&handle, own) == hipSuccess)) {
      mine_ok = 1.0;
    } else {
      (void)hipFree(own);
      own = nullptr;
    }
  }
  (void)hipGetLastError();
  h.own = own;
  std::vector<double> hsend(kHandleDoubles + 1, 0.0), hrecv;
  memcpy(hsend.data(), &handle, sizeof(handle));
  hsend[kHandleDoubles] = mine_ok;
  if (int e = host_all_gather(c, hsend.data(), kHandleDoubles + 1, hrecv, s)) return e;
  bool all = true;
  for (int r = 0; r < world; ++r) all = all && hrecv[(kHandleDoubles + 1) * (size_t)r + kHandleDoubles] == 1.0;
  double map_ok = all ? 1.0 : 0.0;
  h.mapped.assign(world, nullptr);
  h.base.assign(world, nullptr);
  if (all) {
    for (int r = 0; r < world; ++r) {
      if (r == c->rank) {
        h.base[r] = static_cast<unsigned long long*>(own);
        continue;
      }
      hipIpcMemHandle_t hd;
      memcpy(&hd, &hrecv[(kHandleDoubles + 1) * (size_t)r], sizeof(hd));
      void* p = nullptr;
      if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) == hipSuccess && p != nullptr) {
        h.mapped[r] = p;
        h.base[r] = static_cast<unsigned long long*>(p);
        // a first write through the copy path of the runtime: an unreachable mapping is an error code HERE, not a fault
        // of the kernel that stores into it
        const unsigned long long probe = 0x68616c6full + static_cast<unsigned long long>(c->rank);
        unsigned long long back = 0;
        char* word = static_cast<char*>(p) + (rows * kHaloWords + (size_t)c->rank) * sizeof(unsigned long long);
        if (hipMemcpy(word, &probe, sizeof(probe), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(&back, word, sizeof(back), hipMemcpyDeviceToHost) != hipSuccess || back != probe) {
          (void)hipGetLastError();
          map_ok = 0.0;
        }
      } else {
        (void)hipGetLastError();
        map_ok = 0.0;
      }
    }
  }
  std::vector<double> oks;
  if (int e = host_all_gather(c, &map_ok, 1, oks, s)) return e;
  for (double v : oks) all = all && v == 1.0;
  if (!all) {
    halo_ipc_close(c);
    return MHIP_SUCCESS;
  }
  h.capacity = rows;
  h.open = true;
  return MHIP_SUCCESS;
}
// Exchange numbers of the inbox halo: a solve uses seq_base .. seq_base + iterations + 1.  The tag is 32 bits and 0 is
// the tag of a cleared inbox, so the numbers wrap to 1, never through 0 (a re-plan zero-fills the inboxes: after ~4e9
// exchanges -- days at 770 iterations a step -- a number that came round to 0 would have matched them).
static uint32_t next_seq_base(uint32_t base, unsigned used) {
  const unsigned long long next = (unsigned long long)base + used + 2ull;
  return next > 0xf0000000ull ? 1u : (uint32_t)next;   // (a solve never has 2^28 iterations: required where it starts)
}
// Re-agreement of the exchange numbers of the mailbox and the inboxes.  Collective; part of every ghost plan.  A solve
// that ended in an error on some rank (a peer's words that never came, a launch error) may have left the ranks with
// different counts; all of them take the maximum -- ahead of every number any rank has posted or waited for -- and the
// sticky error flag is cleared.  Costs one small all-gather per neighbour-list rebuild.
static int comm_resync(mhip_comm* c, hipStream_t s) {
  unsigned long long st[2] = {0, 0};   // [0] sticky error flag, [1] mailbox exchanges made (device-counted)
  if (c->mbox.status.ptr) {
    MHIP_HIP(hipMemcpyAsync(st, c->mbox.status.ptr, sizeof(st), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
  }
  const double mine[3] = {(double)c->hipc.seq_base, (double)st[1], (double)st[0]};
  std::vector<double> all;
  if (int e = host_all_gather(c, mine, 3, all, s)) return e;
  double seq = 0.0, count = 0.0, flagged = 0.0;
  bool same = true;
  for (int r = 0; r < c->world; ++r) {
    same = same && all[3 * (size_t)r] == mine[0] && all[3 * (size_t)r + 1] == mine[1];
    seq = std::max(seq, all[3 * (size_t)r]);
    count = std::max(count, all[3 * (size_t)r + 1]);
    flagged = std::max(flagged, all[3 * (size_t)r + 2]);
  }
  if (same && flagged == 0.0) return MHIP_SUCCESS;   // (every rebuild of a healthy run)
  c->hipc.seq_base = (uint32_t)seq;
  if (c->mbox.status.ptr) {
    // two beyond the largest count: a rank that posted exchange N + 1 and never finished it has left words tagged
    // N + 1 in its peers' boxes
    const unsigned long long fresh[2] = {0ull, same ? st[1] : (unsigned long long)count + 2ull};
    MHIP_HIP(hipMemcpyAsync(c->mbox.status.ptr, fresh, sizeof(fresh), hipMemcpyHostToDevice, s));
    MHIP_HIP(hipStreamSynchronize(s));
  }
  return MHIP_SUCCESS;
}
// After a ghost plan: room for everybody's rows, and where my rows go in each peer's table.  Collective.
int halo_ipc_plan(mhip_comm* c, const std::vector<size_t>& counts /*[s * W + d]*/, const std::vector<size_t>& owned,
                  hipStream_t s) {
  auto& h = c->hipc;
  auto& gp = c->ghost;
  h.plan_ok = false;
  if (!h.wanted || c->world > kHaloMaxPeers) return comm_resync(c, s);
  const size_t W = (size_t)c->world;
  if (!c->mbox.status.ptr) {
    if (int e = c->mbox.status.reserve(64)) return e;
    MHIP_HIP(hipMemsetAsync(c->mbox.status.ptr, 0, 64, s));
  }
  // rows of every rank's table (the counts matrix and the owned counts are the same on every rank)
  size_t need = 0;
  std::vector<size_t> n_lo(W, 0), n_local(W, 0);
  for (size_t d = 0; d < W; ++d) {
    size_t lo = 0, hi = 0;
    for (size_t p = 0; p < W; ++p)
      if (p != d) (p < d ? lo : hi) += counts[p * W + d];
    n_lo[d] = lo;
    n_local[d] = lo + owned[d] + hi;
    need = std::max(need, n_local[d]);
  }
  bool fresh = false;
  if (!h.open || h.capacity < need) {
    if (int e = halo_ipc_open(c, need + need / 2 + 1024, s)) return e;   // (every rank sees the same `need`)
    if (!h.open) return comm_resync(c, s);
    fresh = true;
  } else if (h.own) {
    // stale words of an earlier plan must never carry a number that comes round again: start from zeros (the
    // collectives of the plan that follow order this before any peer's next push)
    MHIP_HIP(hipMemsetAsync(h.own, 0, h.capacity * kHaloWords * sizeof(unsigned long long), s));
    MHIP_HIP(hipStreamSynchronize(s));
  }
  // nobody pushes before everybody has cleared -- and everybody leaves with the same exchange numbers (comm_resync)
  if (int e = comm_resync(c, s)) return e;
  // where the block of rank R starts in the table of rank d: ghosts of lower ranks in rank order, the owned block,
  // ghosts of higher ranks in rank order (mhip_ghost_layout_from_counts)
  const size_t R = (size_t)c->rank;
  h.dst_first_row.assign(gp.send_peer.size(), 0);
  for (size_t k = 0; k < gp.send_peer.size(); ++k) {
    const size_t d = (size_t)gp.send_peer[k];
    size_t row = 0;
    if (R < d) {
      for (size_t p = 0; p < R; ++p) row += counts[p * W + d];
    } else {
      row = n_lo[d] + owned[d];
      for (size_t p = d + 1; p < R; ++p) row += counts[p * W + d];
    }
    h.dst_first_row[k] = row;
  }
  if (fresh) {
    // newly mapped inboxes: the thing itself once, over the wire the solve will use, before a solve depends on it --
    // every rank pushes rows that name it, collects its ghost rows and checks who they came from; anything short of
    // success on every rank leaves everybody on send / recv
    const size_t nl = gp.n_lo + gp.n + gp.n_hi, ng = gp.n_lo + gp.n_hi;
    // (whatever fails on this rank -- an allocation, a copy, a launch -- is a failed trial, never an early return: the
    //  peers are on their way into the all-gather of the outcomes below and must find this rank there)
    DeviceBuffer tv;
    std::vector<double> hv(6 * nl, -1.0);
    for (size_t r = gp.n_lo; r < gp.n_lo + gp.n; ++r)
      for (int k = 0; k < 6; ++k) hv[6 * r + k] = 1000.0 * (double)(c->rank + 1) + 0.125 * k;
    unsigned long long* stw = c->mbox.status.as<unsigned long long>();
    auto trial = [&]() -> int {
      if (int e = tv.reserve((6 * nl + 8) * sizeof(double))) return e;
      if (nl) MHIP_HIP(hipMemcpyAsync(tv.ptr, hv.data(), 6 * nl * sizeof(double), hipMemcpyHostToDevice, s));
      MHIP_HIP(hipMemsetAsync(stw + 3, 0, 2 * sizeof(unsigned long long), s));   // words 3, 4: a zero `done`, a zero `flips`
      HaloPushArgs push{};
      push.npeers = (int)gp.send_peer.size();
      size_t off = 0;
      for (size_t k = 0; k < gp.send_peer.size(); ++k) {
        push.base[k] = h.base[(size_t)gp.send_peer[k]];
        push.first[k] = (unsigned)off;
        push.dst_row[k] = (unsigned)h.dst_first_row[k];
        off += gp.send_rows[k];
      }
      push.first[gp.send_peer.size()] = (unsigned)off;
      const unsigned* zflips = reinterpret_cast<const unsigned*>(stw + 4);
      const int* zdone = reinterpret_cast<const int*>(stw + 3);
      if (gp.total_send) {
        k_halo_push<<<grid_for(gp.total_send * kHaloWords), kBlock, 0, s>>>(
            push, gp.total_send, gp.send_index_local.as<int32_t>(), tv.as<double>(), h.seq_base, 1, zflips, zdone);
        MHIP_LAUNCH_CHECK();
      }
      if (ng) {
        k_halo_collect<<<grid_for(ng * kHaloWords), kBlock, 0, s>>>(h.base[(size_t)c->rank], gp.n_lo, gp.n, ng,
                                                                   tv.as<double>(), h.seq_base, 1, zflips, zdone, stw,
                                                                   c->timeout_ticks / 4);
        MHIP_LAUNCH_CHECK();
      }
      if (nl) MHIP_HIP(hipMemcpyAsync(hv.data(), tv.ptr, 6 * nl * sizeof(double), hipMemcpyDeviceToHost, s));
      unsigned long long bad = 0;
      MHIP_HIP(hipMemcpyAsync(&bad, stw, sizeof(bad), hipMemcpyDeviceToHost, s));
      MHIP_HIP(hipStreamSynchronize(s));
      if (bad) {
        (void)hipMemsetAsync(stw, 0, sizeof(unsigned long long), s);   // (the sticky flag of this trial)
        return fail(MHIP_ERR_RUNTIME, "trial exchange through the inboxes: status %llu", bad);
      }
      for (size_t k = 0; k < gp.recv_peer.size(); ++k)
        for (size_t r = gp.recv_first_row[k]; r < gp.recv_first_row[k] + gp.recv_rows[k]; ++r)
          for (int q = 0; q < 6; ++q)
            if (hv[6 * r + q] != 1000.0 * (double)(gp.recv_peer[k] + 1) + 0.125 * q)
              return fail(MHIP_ERR_RUNTIME, "trial exchange through the inboxes: row %zu did not come from rank %d", r,
                          gp.recv_peer[k]);
      return MHIP_SUCCESS;
    };
    double trial_ok = (trial() == MHIP_SUCCESS) ? 1.0 : 0.0;
    if (trial_ok == 0.0) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(s);
    }
    h.seq_base = next_seq_base(h.seq_base, 0u);
    std::vector<double> oks;
    if (int e = host_all_gather(c, &trial_ok, 1, oks, s)) return e;
    bool all = true;
    for (double v : oks) all = all && v == 1.0;
    if (!all) {
      halo_ipc_close(c);
      h.wanted = false;   // (not tried again on this communicator)
      return MHIP_SUCCESS;
    }
  }
  h.plan_ok = true;
  return MHIP_SUCCESS;
}

MailboxArgs mailbox_next(mhip_comm* c, int width, double* gathered) {
  MailboxArgs m;
  m.peers = c->mbox.peers.as<unsigned long long*>();
  m.world = c->world;
  m.rank = c->rank;
  m.width = width;
  m.gathered = gathered;
  m.status = c->mbox.status.as<unsigned long long>();
  m.timeout = c->timeout_ticks;
  return m;
}
int mailbox_exchange(mhip_comm* c, int width, const double* local, double* gathered, hipStream_t s) {
  k_mailbox_exchange<<<1, 64, 0, s>>>(mailbox_next(c, width, gathered), local);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}
// after a stream synchronisation: did an exchange time out?
int mailbox_check(mhip_comm* c, hipStream_t s) {
  unsigned long long st[2] = {0, 0};
  MHIP_HIP(hipMemcpyAsync(st, c->mbox.status.ptr, sizeof(st), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  MHIP_REQUIRE(st[0] != 2, MHIP_ERR_RUNTIME,
               "rank %d: a ghost row of a LATER exchange was found in the inbox before this one had been collected -- the "
               "ranks disagree on the exchange numbers (an earlier solve failed on some rank?): run mhip_ghost_plan, "
               "which re-agrees them, before the next solve", c->rank);
  MHIP_REQUIRE(st[0] == 0, MHIP_ERR_RUNTIME,
               "rank %d: a peer's words (reduction record or ghost rows) did not arrive within %.0f s (mailbox exchange "
               "%llu); mhip_ghost_plan re-agrees the exchange numbers before the next solve", c->rank,
               (double)c->timeout_ticks * 1e-8, st[1] + 1);
  return MHIP_SUCCESS;
}
void mailbox_close(mhip_comm* c) {
  for (void* p : c->mbox.mapped)
    if (p) (void)hipIpcCloseMemHandle(p);
  if (c->mbox.own) (void)hipFree(c->mbox.own);
  c->mbox.peers.release();
  c->mbox.status.release();
  c->mbox = mhip_comm::Mailbox{};
}

}  // namespace mhip

extern "C" {

/* Opens the mailbox of a communicator whose ranks all run on this node (see mundy_hip.h).  Collective. */
int mhip_comm_mailbox_open(mhip_comm_t c, int* opened, mhip_stream_t stream) {
  MHIP_REQUIRE(c != nullptr && opened != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  *opened = 0;
  if (c->mbox.open || c->mbox.own) mailbox_close(c);
  hipStream_t s = as_stream(stream);
  const int world = c->world;
  // the slots, then one scratch word per rank (written through the copy engine when the box is mapped: see below)
  const size_t slot_bytes = 2 * (size_t)world * kSlotWords * sizeof(unsigned long long);
  const size_t bytes = slot_bytes + (size_t)world * sizeof(unsigned long long);
  constexpr size_t kHandleDoubles = (sizeof(hipIpcMemHandle_t) + 7) / 8;
  // this rank's box: fine-grained device memory (coherent for the peers that write into it while a kernel polls it)
  double mine_ok = 0.0;
  hipIpcMemHandle_t handle;
  memset(&handle, 0, sizeof(handle));
  void* own = nullptr;
  if (hipExtMallocWithFlags(&own, bytes, hipDeviceMallocFinegrained) == hipSuccess && own != nullptr) {
    if (hipMemsetAsync(own, 0, bytes, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess &&
        (world == 1 || hipIpcGetMemHandle(&handle, own) == hipSuccess)) {
      mine_ok = 1.0;
    } else {
      (void)hipFree(own);
      own = nullptr;
    }
  }
  (void)hipGetLastError();
  c->mbox.own = own;
  // the handles travel through the ordinary transport (bytes in doubles: copied, never computed with)
  if (int e = c->triples.reserve((kHandleDoubles + 8) * (size_t)(world + 1) * sizeof(double))) return e;
  double* send = c->triples.as<double>();
  double* recv = send + kHandleDoubles + 8;
  std::vector<double> hsend(kHandleDoubles + 1, 0.0), hrecv((kHandleDoubles + 1) * (size_t)world, 0.0);
  memcpy(hsend.data(), &handle, sizeof(handle));
  hsend[kHandleDoubles] = mine_ok;
  MHIP_HIP(hipMemcpyAsync(send, hsend.data(), hsend.size() * sizeof(double), hipMemcpyHostToDevice, s));
  if (int e = mhip_comm_all_gather(c, send, kHandleDoubles + 1, recv, stream)) return e;
  MHIP_HIP(hipMemcpyAsync(hrecv.data(), recv, hrecv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  bool all = true;
  for (int r = 0; r < world; ++r) all = all && hrecv[(kHandleDoubles + 1) * (size_t)r + kHandleDoubles] == 1.0;
  // map everybody's box (everybody tries, then everybody agrees: all calls below are made by every rank)
  double map_ok = all ? 1.0 : 0.0;
  c->mbox.mapped.assign(world, nullptr);
  std::vector<unsigned long long*> peers(world, nullptr);
  if (all) {
    for (int r = 0; r < world; ++r) {
      if (r == c->rank) {
        peers[r] = static_cast<unsigned long long*>(own);
        continue;
      }
      hipIpcMemHandle_t h;
      memcpy(&h, &hrecv[(kHandleDoubles + 1) * (size_t)r], sizeof(h));
      void* p = nullptr;
      if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess && p != nullptr) {
        c->mbox.mapped[r] = p;
        peers[r] = static_cast<unsigned long long*>(p);
        // a first write through the runtime's copy path: a mapping this device cannot reach fails HERE, as an error
        // code, not later as a fault of the kernel that stores into it
        const unsigned long long probe = 0x6d626f78ull + static_cast<unsigned long long>(c->rank);
        unsigned long long back = 0;
        char* word = static_cast<char*>(p) + slot_bytes + (size_t)c->rank * sizeof(unsigned long long);
        if (hipMemcpy(word, &probe, sizeof(probe), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(&back, word, sizeof(back), hipMemcpyDeviceToHost) != hipSuccess || back != probe) {
          (void)hipGetLastError();
          map_ok = 0.0;
        }
      } else {
        (void)hipGetLastError();
        map_ok = 0.0;
      }
    }
    if (c->mbox.peers.reserve(world * sizeof(void*)) != MHIP_SUCCESS || c->mbox.status.reserve(64) != MHIP_SUCCESS ||
        hipMemcpyAsync(c->mbox.peers.ptr, peers.data(), world * sizeof(void*), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(c->mbox.status.ptr, 0, 64, s) != hipSuccess)
      map_ok = 0.0;
  }
  double* flag = send;
  double* flags = recv;
  auto agree = [&](double ok, bool* everybody) -> int {
    MHIP_HIP(hipMemcpyAsync(flag, &ok, sizeof(double), hipMemcpyHostToDevice, s));
    if (int e = mhip_comm_all_gather(c, flag, 1, flags, stream)) return e;
    std::vector<double> h(world);
    MHIP_HIP(hipMemcpyAsync(h.data(), flags, world * sizeof(double), hipMemcpyDeviceToHost, s));
    MHIP_HIP(hipStreamSynchronize(s));
    *everybody = true;
    for (double v : h) *everybody = *everybody && v == 1.0;
    return MHIP_SUCCESS;
  };
  if (int e = agree(map_ok, &all)) return e;
  if (all) {  // the thing itself, twice (both slot sets); the exchange counter starts at 0 (status was just zeroed)
    double trial_ok = 1.0;
    for (int round = 0; round < 2 && trial_ok == 1.0; ++round) {
      const double v = 1000.0 * (round + 1) + c->rank;
      MHIP_HIP(hipMemcpyAsync(flag, &v, sizeof(double), hipMemcpyHostToDevice, s));
      if (int e = mailbox_exchange(c, 1, flag, flags, s)) return e;
      std::vector<double> h(world);
      MHIP_HIP(hipMemcpyAsync(h.data(), flags, world * sizeof(double), hipMemcpyDeviceToHost, s));
      MHIP_HIP(hipStreamSynchronize(s));
      for (int r = 0; r < world; ++r)
        if (h[r] != 1000.0 * (round + 1) + r) trial_ok = 0.0;
    }
    if (int e = agree(trial_ok, &all)) return e;
  }
  if (all) {
    c->mbox.open = true;
    *opened = 1;
  } else {
    mailbox_close(c);
  }
  return MHIP_SUCCESS;
}

int mhip_comm_mailbox_close(mhip_comm_t c) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  mailbox_close(c);
  return MHIP_SUCCESS;
}

int mhip_comm_halo_ipc_enable(mhip_comm_t c, int on) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  c->hipc.wanted = on != 0;
  if (!on) halo_ipc_close(c);
  return MHIP_SUCCESS;
}
int mhip_comm_halo_ipc_active(mhip_comm_t c, int* active) {
  MHIP_REQUIRE(c != nullptr && active != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  *active = (c->hipc.open && c->hipc.plan_ok) ? 1 : 0;
  return MHIP_SUCCESS;
}

int mhip_comm_unique_id(unsigned char* id) {
  MHIP_REQUIRE(id != nullptr, MHIP_ERR_INVALID_ARGUMENT, "id is null");
  static_assert(sizeof(ncclUniqueId) == MHIP_COMM_ID_BYTES, "unique id size");
  MHIP_REQUIRE(rccl().ok(), MHIP_ERR_RUNTIME, "librccl.so.1 could not be loaded: %s", dlerror());
  ncclUniqueId u;
  MHIP_RCCL(rccl().GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return MHIP_SUCCESS;
}

int mhip_comm_create_rccl(mhip_comm_t* comm, const unsigned char* id, int rank, int world) {
  MHIP_REQUIRE(comm != nullptr && id != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, MHIP_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, world);
  MHIP_REQUIRE(rccl().ok(), MHIP_ERR_RUNTIME, "librccl.so.1 could not be loaded: %s", dlerror());
  mhip_comm* c = new mhip_comm();
  c->rank = rank;
  c->world = world;
  c->is_rccl = true;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = rccl().CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(MHIP_ERR_RUNTIME, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (e != hipSuccess) {
    mhip_comm_destroy(c);
    return fail(MHIP_ERR_HIP, "communicator stream / events: %s", hipGetErrorString(e));
  }
  *comm = c;
  return MHIP_SUCCESS;
}

int mhip_comm_create_host(mhip_comm_t* comm, int rank, int world, mhip_comm_exchange_fn exchange,
                          mhip_comm_all_gather_fn all_gather, void* user) {
  MHIP_REQUIRE(comm != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, MHIP_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, world);
  MHIP_REQUIRE(world == 1 || (exchange && all_gather), MHIP_ERR_INVALID_ARGUMENT,
               "a host transport of more than one rank needs both callbacks");
  mhip_comm* c = new mhip_comm();
  c->rank = rank;
  c->world = world;
  c->xfn = exchange;
  c->gfn = all_gather;
  c->user = user;
  *comm = c;
  return MHIP_SUCCESS;
}

int mhip_comm_destroy(mhip_comm_t c) {
  if (!c) return MHIP_SUCCESS;
  for (auto ev : c->events) (void)hipEventDestroy(ev);
  c->send_rows.release();
  c->triples.release();
  c->ghost.send_index.release();
  c->ghost.send_index_local.release();
  c->ghost.stage.release();
  c->ghost.regions.release();
  for (DeviceBuffer* b : {&c->migrate.dest, &c->migrate.sortkey, &c->migrate.order, &c->migrate.hist, &c->migrate.stage})
    b->release();
  mailbox_close(c);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  halo_ipc_close(c);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return MHIP_SUCCESS;
}

int mhip_comm_info(mhip_comm_t c, int* rank, int* world, int* is_rccl) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (is_rccl) *is_rccl = c->is_rccl ? 1 : 0;
  return MHIP_SUCCESS;
}

int mhip_comm_all_gather(mhip_comm_t c, const double* send, size_t count, double* recv, mhip_stream_t stream) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  MHIP_REQUIRE(!c->in_flight, MHIP_ERR_RUNTIME, "an exchange is in flight: call mhip_comm_exchange_finish first");
  if (count == 0) return MHIP_SUCCESS;
  MHIP_REQUIRE(send != nullptr && recv != nullptr, MHIP_ERR_INVALID_ARGUMENT, "all_gather buffers are null");
  hipStream_t s = as_stream(stream);
  if (c->is_rccl) {
    // Collectives go straight onto the caller's stream (no hand-over: this one sits on the critical path of every
    // iteration).  RCCL orders the launches of one communicator itself when the stream changes between calls, and
    // the caller has already waited for the last exchange (in_flight is false), so nothing is serialised needlessly.
    MHIP_RCCL(rccl().AllGather(send, recv, count, ncclDouble, c->nccl, s));
    return MHIP_SUCCESS;
  }
  if (c->world == 1) {
    MHIP_HIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return MHIP_SUCCESS;
  }
  MHIP_HIP(hipStreamSynchronize(s));
  const int e = c->gfn(c->user, send, count, recv);
  MHIP_REQUIRE(e == 0, MHIP_ERR_RUNTIME, "the host all_gather callback returned %d", e);
  return MHIP_SUCCESS;
}

int mhip_comm_exchange_start(mhip_comm_t c, int nsend, const int* send_peer, const double* const* send_buf,
                             const size_t* send_count, int nrecv, const int* recv_peer, double* const* recv_buf,
                             const size_t* recv_count, mhip_stream_t stream) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  MHIP_REQUIRE(!c->in_flight, MHIP_ERR_RUNTIME, "an exchange is already in flight");
  MHIP_REQUIRE(nsend >= 0 && nrecv >= 0, MHIP_ERR_INVALID_ARGUMENT, "negative message count");
  MHIP_REQUIRE(nsend == 0 || (send_peer && send_buf && send_count), MHIP_ERR_INVALID_ARGUMENT, "send lists are null");
  MHIP_REQUIRE(nrecv == 0 || (recv_peer && recv_buf && recv_count), MHIP_ERR_INVALID_ARGUMENT, "recv lists are null");
  for (int k = 0; k < nsend; ++k) {
    // RCCL delivers a message to oneself as a local copy; the host transport does not loop back
    MHIP_REQUIRE(send_peer[k] >= 0 && send_peer[k] < c->world && (c->is_rccl || send_peer[k] != c->rank),
                 MHIP_ERR_INVALID_ARGUMENT, "send peer %d is not another rank of this %d-rank group", send_peer[k],
                 c->world);
    MHIP_REQUIRE(send_count[k] == 0 || send_buf[k], MHIP_ERR_INVALID_ARGUMENT, "send buffer %d is null", k);
  }
  for (int k = 0; k < nrecv; ++k) {
    MHIP_REQUIRE(recv_peer[k] >= 0 && recv_peer[k] < c->world && (c->is_rccl || recv_peer[k] != c->rank),
                 MHIP_ERR_INVALID_ARGUMENT, "recv peer %d is not another rank of this %d-rank group", recv_peer[k],
                 c->world);
    MHIP_REQUIRE(recv_count[k] == 0 || recv_buf[k], MHIP_ERR_INVALID_ARGUMENT, "recv buffer %d is null", k);
  }
  hipStream_t s = as_stream(stream);
  if (c->is_rccl) {
    MHIP_HIP(hipEventRecord(c->ready, s));
    MHIP_HIP(hipStreamWaitEvent(c->comm_stream, c->ready, 0));
    MHIP_RCCL(rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < nrecv && bad == ncclSuccess; ++k)
      if (recv_count[k]) bad = rccl().Recv(recv_buf[k], recv_count[k], ncclDouble, recv_peer[k], c->nccl, c->comm_stream);
    for (int k = 0; k < nsend && bad == ncclSuccess; ++k)
      if (send_count[k]) bad = rccl().Send(send_buf[k], send_count[k], ncclDouble, send_peer[k], c->nccl, c->comm_stream);
    ncclResult_t end = rccl().GroupEnd();  // always closes the group, even after a failed enqueue
    MHIP_RCCL(bad);
    MHIP_RCCL(end);
    MHIP_HIP(hipEventRecord(c->done, c->comm_stream));
    c->in_flight = true;
    return MHIP_SUCCESS;
  }
  c->send_peer.assign(send_peer, send_peer + nsend);
  c->send_buf.assign(send_buf, send_buf + nsend);
  c->send_count.assign(send_count, send_count + nsend);
  c->recv_peer.assign(recv_peer, recv_peer + nrecv);
  c->recv_buf.assign(recv_buf, recv_buf + nrecv);
  c->recv_count.assign(recv_count, recv_count + nrecv);
  c->in_flight = true;
  return MHIP_SUCCESS;
}

int mhip_comm_exchange_finish(mhip_comm_t c, mhip_stream_t stream) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  MHIP_REQUIRE(c->in_flight, MHIP_ERR_RUNTIME, "no exchange in flight");
  c->in_flight = false;
  hipStream_t s = as_stream(stream);
  if (c->is_rccl) {
    MHIP_HIP(hipStreamWaitEvent(s, c->done, 0));
    return MHIP_SUCCESS;
  }
  if (c->world == 1 || (c->send_peer.empty() && c->recv_peer.empty())) return MHIP_SUCCESS;
  MHIP_HIP(hipStreamSynchronize(s));
  const int e = c->xfn(c->user, (int)c->send_peer.size(), c->send_peer.data(), c->send_buf.data(), c->send_count.data(),
                       (int)c->recv_peer.size(), c->recv_peer.data(), c->recv_buf.data(), c->recv_count.data());
  MHIP_REQUIRE(e == 0, MHIP_ERR_RUNTIME, "the host exchange callback returned %d", e);
  return MHIP_SUCCESS;
}

int mhip_ghost_layout_from_counts(int world, int rank, size_t n_owned, const size_t* counts, size_t* num_ghost_lo,
                                  size_t* num_ghost_hi, int* num_send, int* send_peer, size_t* send_rows, int* num_recv,
                                  int* recv_peer, size_t* recv_first_row, size_t* recv_rows) {
  MHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, MHIP_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, world);
  MHIP_REQUIRE(counts && num_ghost_lo && num_ghost_hi && num_send && num_recv, MHIP_ERR_INVALID_ARGUMENT,
               "null argument");
  MHIP_REQUIRE(world == 1 || (send_peer && send_rows && recv_peer && recv_first_row && recv_rows),
               MHIP_ERR_INVALID_ARGUMENT, "peer lists are null");
  const size_t W = (size_t)world, R = (size_t)rank;
  size_t n_lo = 0, n_hi = 0;
  int ns = 0, nr = 0;
  for (size_t p = 0; p < W; ++p) {
    if (p == R) continue;
    if (const size_t sc = counts[R * W + p]) {
      send_peer[ns] = (int)p;
      send_rows[ns++] = sc;
    }
    (p < R ? n_lo : n_hi) += counts[p * W + R];
  }
  size_t row = 0;
  for (size_t p = 0; p < W; ++p) {
    if (p == R) {
      row = n_lo + n_owned;  // ghosts of higher ranks sit after the owned block
      continue;
    }
    const size_t rc = counts[p * W + R];
    if (rc) {
      recv_peer[nr] = (int)p;
      recv_first_row[nr] = row;
      recv_rows[nr++] = rc;
    }
    row += rc;
  }
  *num_ghost_lo = n_lo;
  *num_ghost_hi = n_hi;
  *num_send = ns;
  *num_recv = nr;
  return MHIP_SUCCESS;
}

int mhip_ghost_plan(mhip_comm_t c, size_t n, const double* aabb, double buffer, mhip_ghost_layout* layout,
                    mhip_stream_t stream) {
  TraceRange trace_range("ghost plan (coarse_search(comm) + change_ghosting)");
  MHIP_REQUIRE(c != nullptr && layout != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(n == 0 || aabb != nullptr, MHIP_ERR_INVALID_ARGUMENT, "aabb is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  hipStream_t s = as_stream(stream);
  auto& gp = c->ghost;
  gp.valid = false;
  const int W = c->world, R = c->rank;
  // rank regions: kGhostBoxes chunk boxes + their union per rank, all-gathered device to device
  constexpr int kGhostBoxes = 64;
  const size_t region = 6 * (size_t)(kGhostBoxes + 1);
  if (int e = gp.regions.reserve((region + (size_t)W * region + 2) * sizeof(double))) return e;
  double* mine = gp.regions.as<double>();
  double* all = mine + region;
  if (int e = mhip_aabb_chunk_bounds(n, aabb, buffer, kGhostBoxes, mine, stream)) return e;
  if (int e = mhip_comm_all_gather(c, mine, region, all, stream)) return e;
  // per peer: the owned bodies whose grown box meets any of the peer's boxes
  if (int e = gp.send_index.reserve(((size_t)(W > 1 ? W - 1 : 1) * n + 2) * sizeof(int32_t))) return e;
  std::vector<double> send_cnt((size_t)W + 1, 0.0);   // [W] = the bodies this rank owns
  send_cnt[(size_t)W] = (double)n;
  size_t off = 0;
  for (int p = 0; p < W; ++p) {
    if (p == R || n == 0) continue;
    size_t cnt = 0;
    if (int e = mhip_select_aabb_overlap_any(n, aabb, buffer, kGhostBoxes, all + (size_t)p * region,
                                             gp.send_index.as<int32_t>() + off, &cnt, stream))
      return e;
    send_cnt[(size_t)p] = (double)cnt;
    off += cnt;
  }
  gp.total_send = off;
  std::vector<double> gathered_counts;  // per rank: its W send counts, then its owned count
  if (int e = host_all_gather(c, send_cnt.data(), (size_t)W + 1, gathered_counts, s)) return e;
  std::vector<size_t> cm((size_t)W * W), owned((size_t)W);  // cm[s][d] = bodies rank s sends to rank d
  for (size_t r = 0; r < (size_t)W; ++r) {
    for (size_t d = 0; d < (size_t)W; ++d) cm[r * W + d] = (size_t)gathered_counts[r * (W + 1) + d];
    owned[r] = (size_t)gathered_counts[r * (W + 1) + W];
  }
  gp.n = n;
  {
    gp.send_peer.assign((size_t)W, 0); gp.send_rows.assign((size_t)W, 0);
    gp.recv_peer.assign((size_t)W, 0); gp.recv_first_row.assign((size_t)W, 0); gp.recv_rows.assign((size_t)W, 0);
    int ns = 0, nr = 0;
    if (int e = mhip_ghost_layout_from_counts(W, R, n, cm.data(), &gp.n_lo, &gp.n_hi, &ns, gp.send_peer.data(),
                                              gp.send_rows.data(), &nr, gp.recv_peer.data(), gp.recv_first_row.data(),
                                              gp.recv_rows.data()))
      return e;
    gp.send_peer.resize((size_t)ns); gp.send_rows.resize((size_t)ns);
    gp.recv_peer.resize((size_t)nr); gp.recv_first_row.resize((size_t)nr); gp.recv_rows.resize((size_t)nr);
  }
  MHIP_REQUIRE(gp.n_lo + n + gp.n_hi < (1u << 31), MHIP_ERR_RUNTIME, "too many local bodies");
  if (int e = gp.send_index_local.reserve((gp.total_send + 2) * sizeof(int32_t))) return e;
  if (gp.total_send) {
    k_offset_i32<<<grid_for(gp.total_send), kBlock, 0, s>>>(gp.total_send, gp.send_index.as<int32_t>(), (int32_t)gp.n_lo,
                                                           gp.send_index_local.as<int32_t>());
    MHIP_LAUNCH_CHECK();
  }
  gp.valid = true;
  if (int e = halo_ipc_plan(c, cm, owned, s)) return e;
  layout->num_ghost_lo = gp.n_lo;
  layout->num_owned = n;
  layout->num_ghost_hi = gp.n_hi;
  layout->num_sent = gp.total_send;
  layout->halo = mhip_velocity_halo{nullptr,
                                    (int)gp.send_peer.size(), gp.send_peer.data(), gp.send_rows.data(),
                                    gp.send_index_local.as<int32_t>(),
                                    (int)gp.recv_peer.size(), gp.recv_peer.data(), gp.recv_first_row.data(),
                                    gp.recv_rows.data()};
  return MHIP_SUCCESS;
}

int mhip_ghost_exchange(mhip_comm_t c, size_t width, const double* records, double* local, mhip_stream_t stream) {
  TraceRange trace_range("ghost exchange");
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  auto& gp = c->ghost;
  MHIP_REQUIRE(gp.valid, MHIP_ERR_RUNTIME, "mhip_ghost_plan has not been called");
  MHIP_REQUIRE(width >= 1, MHIP_ERR_INVALID_ARGUMENT, "width must be at least 1");
  const size_t n_local = gp.n_lo + gp.n + gp.n_hi;
  MHIP_REQUIRE(n_local == 0 || local != nullptr, MHIP_ERR_INVALID_ARGUMENT, "local is null");
  MHIP_REQUIRE(gp.n == 0 || records != nullptr, MHIP_ERR_INVALID_ARGUMENT, "records is null");
  hipStream_t s = as_stream(stream);
  if (gp.n)
    MHIP_HIP(hipMemcpyAsync(local + gp.n_lo * width, records, gp.n * width * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (gp.send_peer.empty() && gp.recv_peer.empty()) return MHIP_SUCCESS;
  if (int e = gp.stage.reserve((gp.total_send * width + 2) * sizeof(double))) return e;
  double* packed = gp.stage.as<double>();
  if (gp.total_send)
    if (int e = mhip_gather_rows(gp.total_send, width, gp.send_index.as<int32_t>(), records, packed, stream)) return e;
  std::vector<const double*> sbuf(gp.send_peer.size());
  std::vector<size_t> scount(gp.send_peer.size());
  size_t off = 0;
  for (size_t k = 0; k < gp.send_peer.size(); ++k) {
    sbuf[k] = packed + off * width;
    scount[k] = gp.send_rows[k] * width;
    off += gp.send_rows[k];
  }
  std::vector<double*> rbuf(gp.recv_peer.size());
  std::vector<size_t> rcount(gp.recv_peer.size());
  for (size_t k = 0; k < gp.recv_peer.size(); ++k) {
    rbuf[k] = local + gp.recv_first_row[k] * width;
    rcount[k] = gp.recv_rows[k] * width;
  }
  if (int e = mhip_comm_exchange_start(c, (int)sbuf.size(), gp.send_peer.data(), sbuf.data(), scount.data(),
                                       (int)rbuf.size(), gp.recv_peer.data(), rbuf.data(), rcount.data(), stream))
    return e;
  return mhip_comm_exchange_finish(c, stream);
}

/* Lattice points of a (2^level)^3 cube in the visiting order of mundy::math::hilbert_3d (mundy/math/src/mundy_math/
 * Hilbert.hpp:48-83) started at the origin with axes (x, y, z): each state (corner, three signed axes) expands into its
 * eight children in curve order.  table[ix][iy][iz] = position of the cell along the curve.  Host only. */
int mhip_hilbert_key_table(int level, int32_t* table) {
  MHIP_REQUIRE(table != nullptr, MHIP_ERR_INVALID_ARGUMENT, "table is null");
  MHIP_REQUIRE(level >= 0 && level <= 8, MHIP_ERR_INVALID_ARGUMENT, "level %d is not in 0..8", level);
  struct State {
    int c[3];
    int d[3][3];  // dr1, dr2, dr3
  };
  std::vector<State> cur(1), next;
  cur[0] = State{{0, 0, 0}, {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  for (int s = 1 << level; s > 1; s /= 2) {
    const int h = s / 2;
    next.clear();
    next.reserve(cur.size() * 8);
    for (const State& st : cur) {
      int cn[3];  // the corner moved to the low end of every axis that points backwards
      for (int a = 0; a < 3; ++a) {
        cn[a] = st.c[a];
        for (int k = 0; k < 3; ++k)
          if (st.d[k][a] < 0) cn[a] -= h * st.d[k][a];
      }
      const int (&d1)[3] = st.d[0];
      const int (&d2)[3] = st.d[1];
      const int (&d3)[3] = st.d[2];
      auto child = [&](int k1, int k2, int k3, const int* a1, int s1, const int* a2, int s2, const int* a3, int s3) {
        State ch;
        for (int a = 0; a < 3; ++a) {
          ch.c[a] = cn[a] + h * (k1 * d1[a] + k2 * d2[a] + k3 * d3[a]);
          ch.d[0][a] = s1 * a1[a];
          ch.d[1][a] = s2 * a2[a];
          ch.d[2][a] = s3 * a3[a];
        }
        next.push_back(ch);
      };
      child(0, 0, 0, d2, 1, d3, 1, d1, 1);
      child(1, 0, 0, d3, 1, d1, 1, d2, 1);
      child(1, 1, 0, d3, 1, d1, 1, d2, 1);
      child(0, 1, 0, d1, -1, d2, -1, d3, 1);
      child(0, 1, 1, d1, -1, d2, -1, d3, 1);
      child(1, 1, 1, d3, -1, d1, 1, d2, -1);
      child(1, 0, 1, d3, -1, d1, 1, d2, -1);
      child(0, 0, 1, d2, 1, d3, -1, d1, -1);
    }
    cur.swap(next);
  }
  const size_t n = static_cast<size_t>(1) << level;
  for (size_t k = 0; k < cur.size(); ++k) {
    const State& st = cur[k];
    MHIP_REQUIRE(st.c[0] >= 0 && st.c[1] >= 0 && st.c[2] >= 0 && (size_t)st.c[0] < n && (size_t)st.c[1] < n &&
                     (size_t)st.c[2] < n,
                 MHIP_ERR_RUNTIME, "curve left the lattice");
    table[((size_t)st.c[0] * n + (size_t)st.c[1]) * n + (size_t)st.c[2]] = static_cast<int32_t>(k);
  }
  return MHIP_SUCCESS;
}

int mhip_curve_cut(mhip_comm_t c, size_t n, const uint32_t* keys, const double* weights, size_t ncell,
                   int64_t* splitters, mhip_stream_t stream) {
  TraceRange trace_range("curve cut by work (stk::balance)");
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  MHIP_REQUIRE(n == 0 || keys != nullptr, MHIP_ERR_INVALID_ARGUMENT, "keys is null");
  MHIP_REQUIRE(ncell >= 1 && ncell <= (1u << 24), MHIP_ERR_INVALID_ARGUMENT, "ncell out of range");
  MHIP_REQUIRE(c->world == 1 || splitters != nullptr, MHIP_ERR_INVALID_ARGUMENT, "splitters is null");
  hipStream_t s = as_stream(stream);
  const size_t W = (size_t)c->world;
  auto& mp = c->migrate;
  if (int e = mp.hist.reserve((ncell + W * ncell + 2) * sizeof(double))) return e;
  double* mine = mp.hist.as<double>();
  double* all = mine + ncell;
  MHIP_HIP(hipMemsetAsync(mine, 0, ncell * sizeof(double), s));
  if (n) {
    k_weighted_hist<<<grid_for(n), kBlock, 0, s>>>(n, keys, weights, ncell, mine);
    MHIP_LAUNCH_CHECK();
  }
  if (int e = mhip_comm_all_gather(c, mine, ncell, all, stream)) return e;
  std::vector<double> host(W * ncell);
  MHIP_HIP(hipMemcpyAsync(host.data(), all, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  // the same sums in the same order on every rank: the same cuts
  std::vector<double> cum(ncell);
  double run = 0.0;
  for (size_t k = 0; k < ncell; ++k) {
    double t = 0.0;
    for (size_t r = 0; r < W; ++r) t += host[r * ncell + k];
    run += t;
    cum[k] = run;
  }
  for (size_t r = 1; r < W; ++r) {  // first cell at which the cumulative weight reaches r / W of the total
    const double target = cum[ncell - 1] * (double)r / (double)W;
    size_t lo = 0, hi = ncell;
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (cum[mid] < target) lo = mid + 1; else hi = mid;
    }
    splitters[r - 1] = static_cast<int64_t>(lo);
  }
  return MHIP_SUCCESS;
}

int mhip_migrate_plan(mhip_comm_t c, size_t n, const uint32_t* keys, const int64_t* splitters, size_t* n_new,
                      size_t* num_sent, size_t* num_received, mhip_stream_t stream) {
  TraceRange trace_range("migration plan");
  MHIP_REQUIRE(c != nullptr && n_new != nullptr, MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(n == 0 || keys != nullptr, MHIP_ERR_INVALID_ARGUMENT, "keys is null");
  MHIP_REQUIRE(c->world == 1 || splitters != nullptr, MHIP_ERR_INVALID_ARGUMENT, "splitters is null");
  MHIP_REQUIRE(n < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies");
  hipStream_t s = as_stream(stream);
  const int W = c->world, R = c->rank;
  auto& mp = c->migrate;
  mp.valid = false;
  for (int r = 1; r + 1 < W; ++r)
    MHIP_REQUIRE(splitters[r - 1] <= splitters[r], MHIP_ERR_INVALID_ARGUMENT, "splitters must not decrease");
  if (int e = mp.dest.reserve(((size_t)W + 2) * sizeof(long long))) return e;
  if (int e = mp.sortkey.reserve((n + 2) * sizeof(unsigned long long))) return e;
  if (int e = mp.order.reserve((n + 2) * sizeof(int32_t))) return e;
  if (int e = mp.hist.reserve(((size_t)W + 2) * sizeof(double))) return e;
  long long* d_split = mp.dest.as<long long>();
  double* d_count = mp.hist.as<double>();
  if (W > 1) MHIP_HIP(hipMemcpyAsync(d_split, splitters, (size_t)(W - 1) * sizeof(long long), hipMemcpyHostToDevice, s));
  MHIP_HIP(hipMemsetAsync(d_count, 0, (size_t)W * sizeof(double), s));
  if (n) {
    k_migrate_dest<<<grid_for(n), kBlock, 0, s>>>(n, keys, W - 1, d_split, mp.sortkey.as<unsigned long long>(), d_count);
    MHIP_LAUNCH_CHECK();
    if (int e = mhip_sort_by_key_u64(n, reinterpret_cast<const uint64_t*>(mp.sortkey.ptr), mp.order.as<int32_t>(), stream))
      return e;
  }
  std::vector<double> mine((size_t)W), matrix;  // matrix[src][dst]
  MHIP_HIP(hipMemcpyAsync(mine.data(), d_count, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, s));
  MHIP_HIP(hipStreamSynchronize(s));
  if (int e = host_all_gather(c, mine.data(), (size_t)W, matrix, s)) return e;
  mp.send_peer.clear(); mp.send_first_row.clear(); mp.send_rows.clear(); mp.recv_peer.clear(); mp.recv_rows.clear();
  size_t row = 0, arriving = 0, leaving = 0;
  for (int p = 0; p < W; ++p) {
    const size_t to_p = (size_t)matrix[(size_t)R * W + p], from_p = (size_t)matrix[(size_t)p * W + R];
    if (p == R) {
      mp.keep = to_p;
      mp.keep_first = row;
    } else {
      if (to_p) {
        mp.send_peer.push_back(p);
        mp.send_first_row.push_back(row);
        mp.send_rows.push_back(to_p);
        leaving += to_p;
      }
      if (from_p) {
        mp.recv_peer.push_back(p);
        mp.recv_rows.push_back(from_p);
        arriving += from_p;
      }
    }
    row += to_p;
  }
  MHIP_REQUIRE(row == n, MHIP_ERR_RUNTIME, "destination counts (%zu) do not add up to the owned bodies (%zu)", row, n);
  mp.n = n;
  mp.n_new = mp.keep + arriving;
  MHIP_REQUIRE(mp.n_new < (1u << 31), MHIP_ERR_RUNTIME, "too many bodies after the migration");
  mp.valid = true;
  *n_new = mp.n_new;
  if (num_sent) *num_sent = leaving;
  if (num_received) *num_received = arriving;
  return MHIP_SUCCESS;
}

int mhip_migrate_exchange(mhip_comm_t c, size_t width, const double* records, double* out, mhip_stream_t stream) {
  TraceRange trace_range("migration exchange");
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  auto& mp = c->migrate;
  MHIP_REQUIRE(mp.valid, MHIP_ERR_RUNTIME, "mhip_migrate_plan has not been called");
  MHIP_REQUIRE(width >= 1, MHIP_ERR_INVALID_ARGUMENT, "width must be at least 1");
  MHIP_REQUIRE(mp.n == 0 || records != nullptr, MHIP_ERR_INVALID_ARGUMENT, "records is null");
  MHIP_REQUIRE(mp.n_new == 0 || out != nullptr, MHIP_ERR_INVALID_ARGUMENT, "out is null");
  MHIP_REQUIRE(out == nullptr || out != records, MHIP_ERR_INVALID_ARGUMENT, "the exchange cannot run in place");
  hipStream_t s = as_stream(stream);
  if (int e = mp.stage.reserve((mp.n * width + 2) * sizeof(double))) return e;
  double* packed = mp.stage.as<double>();  // the owned rows grouped by destination, each group in its present order
  if (mp.n)
    if (int e = mhip_gather_rows(mp.n, width, mp.order.as<int32_t>(), records, packed, stream)) return e;
  if (mp.keep)
    MHIP_HIP(hipMemcpyAsync(out, packed + mp.keep_first * width, mp.keep * width * sizeof(double),
                            hipMemcpyDeviceToDevice, s));
  if (mp.send_peer.empty() && mp.recv_peer.empty()) return MHIP_SUCCESS;
  std::vector<const double*> sbuf(mp.send_peer.size());
  std::vector<size_t> scount(mp.send_peer.size());
  for (size_t k = 0; k < mp.send_peer.size(); ++k) {
    sbuf[k] = packed + mp.send_first_row[k] * width;
    scount[k] = mp.send_rows[k] * width;
  }
  std::vector<double*> rbuf(mp.recv_peer.size());
  std::vector<size_t> rcount(mp.recv_peer.size());
  size_t row = mp.keep;  // arrivals follow the rows that stay, peers in increasing rank
  for (size_t k = 0; k < mp.recv_peer.size(); ++k) {
    rbuf[k] = out + row * width;
    rcount[k] = mp.recv_rows[k] * width;
    row += mp.recv_rows[k];
  }
  if (int e = mhip_comm_exchange_start(c, (int)sbuf.size(), mp.send_peer.data(), sbuf.data(), scount.data(),
                                       (int)rbuf.size(), mp.recv_peer.data(), rbuf.data(), rcount.data(), stream))
    return e;
  return mhip_comm_exchange_finish(c, stream);
}

int mhip_bbpgd_solve_contact_distributed(mhip_contact_op_t op, mhip_comm_t c, const mhip_velocity_halo* halo,
                                         size_t interior_contacts, const double* q, const mhip_space* space,
                                         const mhip_pgd_config* config, double* x, double* g, double* x_tmp,
                                         double* g_tmp, unsigned poll_every, mhip_solve_result* result,
                                         mhip_dist_profile* profile, mhip_stream_t stream) {
  TraceRange trace_range("solve_cqpp (domain-decomposed BBPGD)");
  MHIP_REQUIRE(op != nullptr && c != nullptr && halo != nullptr && result != nullptr && config != nullptr,
               MHIP_ERR_INVALID_ARGUMENT, "null argument");
  MHIP_REQUIRE(halo->num_send_peers >= 0 && halo->num_recv_peers >= 0, MHIP_ERR_INVALID_ARGUMENT,
               "negative peer count");
  *result = mhip_solve_result{};  // (read on every exit path, also those that come before the first poll)
  if (poll_every == 0) poll_every = 64;  // (as the fused driver; 32 cost 2-3 % of a step at 1.25e5 ... 1e6 rods per rank)
  hipStream_t s = as_stream(stream);
  if (c->in_flight)  // an earlier solve ended in an error between start and finish: close that exchange first
    if (int e = mhip_comm_exchange_finish(c, stream)) return e;
  // message lists of the velocity halo (rows of 6 doubles)
  size_t send_total = 0;
  for (int k = 0; k < halo->num_send_peers; ++k) send_total += halo->send_rows[k];
  MHIP_REQUIRE(send_total == 0 || (halo->send_index && halo->velocity), MHIP_ERR_INVALID_ARGUMENT,
               "velocity halo without send_index / velocity");
  if (int e = c->send_rows.reserve((6 * send_total + 2) * sizeof(double))) return e;
  constexpr int kRed = MHIP_BBPGD_REDUCTION_WIDTH;
  if (int e = c->triples.reserve((kRed + kRed * (size_t)c->world) * sizeof(double))) return e;
  double* local3 = c->triples.as<double>();  // this rank's reduction record
  double* gathered = local3 + kRed;
  std::vector<const double*> sbuf(halo->num_send_peers);
  std::vector<size_t> scount(halo->num_send_peers);
  {
    size_t off = 0;
    for (int k = 0; k < halo->num_send_peers; ++k) {
      sbuf[k] = c->send_rows.as<double>() + 6 * off;
      scount[k] = 6 * halo->send_rows[k];
      off += halo->send_rows[k];
    }
  }
  std::vector<double*> rbuf(halo->num_recv_peers);
  std::vector<size_t> rcount(halo->num_recv_peers);
  for (int k = 0; k < halo->num_recv_peers; ++k) {
    MHIP_REQUIRE(halo->recv_rows[k] == 0 || halo->velocity, MHIP_ERR_INVALID_ARGUMENT, "velocity halo without velocity");
    rbuf[k] = halo->velocity + 6 * halo->recv_first_row[k];
    rcount[k] = 6 * halo->recv_rows[k];
  }
  const bool has_halo = c->world > 1 && (halo->num_send_peers > 0 || halo->num_recv_peers > 0);
  // the halo of the current ghost plan can go through the inboxes (same lists, same row numbering)
  auto& hi = c->hipc;
  const bool ipc = has_halo && hi.open && hi.plan_ok && c->ghost.valid && halo->velocity != nullptr &&
                   halo->send_index == c->ghost.send_index_local.as<int32_t>() &&
                   halo->num_send_peers == (int)c->ghost.send_peer.size() && halo->num_send_peers <= kHaloMaxPeers;
  HaloPushArgs push{};
  const unsigned* st_flips = nullptr;
  const int* st_done = nullptr;
  const size_t n_ghost = c->ghost.n_lo + c->ghost.n_hi;
  if (ipc) {
    push.npeers = halo->num_send_peers;
    size_t off = 0;
    for (int k = 0; k < halo->num_send_peers; ++k) {
      push.base[k] = hi.base[(size_t)halo->send_peer[k]];
      push.first[k] = (unsigned)off;
      push.dst_row[k] = (unsigned)hi.dst_first_row[(size_t)k];
      off += halo->send_rows[k];
    }
    push.first[halo->num_send_peers] = (unsigned)off;
  }
  const uint32_t seq_base = hi.seq_base;
  MHIP_REQUIRE(!ipc || config->max_iters < (1u << 28), MHIP_ERR_INVALID_ARGUMENT,
               "max_iters %u: the exchange numbers of the inbox halo allow 2^28 iterations per solve", config->max_iters);
  size_t C = 0;
  if (int e = mhip_contact_op_sizes(op, &C, nullptr)) return e;
  MHIP_REQUIRE(interior_contacts <= C, MHIP_ERR_INVALID_ARGUMENT, "interior_contacts %zu exceeds the %zu constraints",
               interior_contacts, C);
  // The length of the stretches between polls must be the SAME decision on every rank (a rank that polls out of phase
  // stalls the others in the record exchange): it is taken from the global constraint count, not from this rank's
  size_t C_global = C;
  if (c->world > 1) {
    const double mine = (double)C;
    std::vector<double> every;
    if (int e = host_all_gather(c, &mine, 1, every, s)) return e;
    double sum = 0.0;
    for (double v : every) sum += v;
    C_global = (size_t)sum;
  }
  c->last_halo_path = !has_halo ? 0 : (ipc ? 1 : 2);
  c->last_record_path = c->mbox.open ? (c->world <= kMailboxMaxWorld ? 1 : 2) : 3;
  const unsigned fault_at_poll = c->fault_at_poll;
  c->fault_at_poll = 0;

  if (int e = mhip_bbpgd_stage_begin(op, q, space, config, x, g, x_tmp, g_tmp, stream)) return e;
  stage_state_words(op, &st_flips, &st_done);

  // sampled timing: every kStride-th iteration of a chunk is bracketed by events
  constexpr unsigned kStride = 8, kEv = 7;
  const bool prof = profile != nullptr;
  const unsigned slots = (poll_every + kStride - 1) / kStride;
  if (prof && c->events.size() < (size_t)kEv * slots) {
    const size_t old = c->events.size();
    c->events.resize((size_t)kEv * slots);
    for (size_t k = old; k < c->events.size(); ++k) MHIP_HIP(hipEventCreate(&c->events[k]));
  }
  if (prof) {
    *profile = mhip_dist_profile{};
    profile->halo_path = c->last_halo_path;
    profile->record_path = c->last_record_path;
  }

  auto iteration = [&](int init, hipEvent_t* ev) -> int {
    if (ev) MHIP_HIP(hipEventRecord(ev[0], s));
    if (int e = mhip_bbpgd_stage_body(op, init, stream)) return e;
    if (ev) MHIP_HIP(hipEventRecord(ev[1], s));
    if (ipc) {
      if (send_total) {   // owned boundary rows straight into the tables' inboxes of the ranks that hold them as ghosts
        k_halo_push<<<grid_for(send_total * kHaloWords), kBlock, 0, s>>>(push, send_total, halo->send_index,
                                                                        halo->velocity, seq_base, init, st_flips, st_done);
        MHIP_LAUNCH_CHECK();
      }
    } else if (has_halo) {
      if (send_total)
        if (int e = mhip_gather_rows(send_total, 6, halo->send_index, halo->velocity, c->send_rows.as<double>(), stream))
          return e;
      if (int e = mhip_comm_exchange_start(c, halo->num_send_peers, halo->send_peer, sbuf.data(), scount.data(),
                                           halo->num_recv_peers, halo->recv_peer, rbuf.data(), rcount.data(), stream))
        return e;
    }
    if (ev) MHIP_HIP(hipEventRecord(ev[2], s));
    // interior contacts need only this rank's own rows: swept while the ghost rows are in flight
    if (int e = mhip_bbpgd_stage_constraint_range(op, init, 0, interior_contacts, stream)) return e;
    if (ev) MHIP_HIP(hipEventRecord(ev[3], s));
    if (ipc) {
      if (n_ghost) {   // my ghost rows, as they arrive in my inbox, into the velocity table
        k_halo_collect<<<grid_for(n_ghost * kHaloWords), kBlock, 0, s>>>(
            hi.base[(size_t)c->rank], c->ghost.n_lo, c->ghost.n, n_ghost, halo->velocity, seq_base, init, st_flips,
            st_done, c->mbox.status.as<unsigned long long>(), c->timeout_ticks);
        MHIP_LAUNCH_CHECK();
      }
    } else if (has_halo) {
      if (int e = mhip_comm_exchange_finish(c, stream)) return e;
    }
    if (ev) MHIP_HIP(hipEventRecord(ev[4], s));
    if (int e = mhip_bbpgd_stage_constraint_range(op, init, interior_contacts, C - interior_contacts, stream)) return e;
    if (ev) MHIP_HIP(hipEventRecord(ev[5], s));
    // ev[5] .. ev[6]: the reduction of the iteration -- this rank's record, its exchange with every rank (the wait for
    // the slowest of them included) and the finalize
    if (c->mbox.open && c->world <= kMailboxMaxWorld) {
      // the record is formed, posted, everybody's collected and the iteration finalized in one launch
      if (int e = stage_reduce_exchange_finalize(op, init, mailbox_next(c, kRed, gathered), s)) return e;
      if (ev) MHIP_HIP(hipEventRecord(ev[6], s));
      return MHIP_SUCCESS;
    }
    if (c->mbox.open) {  // the record is posted, and everybody's collected, by the kernel that forms it
      if (int e = stage_reduce_exchange(op, init, local3, mailbox_next(c, kRed, gathered), s)) return e;
    } else {
      if (int e = mhip_bbpgd_stage_reduce(op, init, local3, stream)) return e;
      if (int e = mhip_comm_all_gather(c, local3, kRed, gathered, stream)) return e;
    }
    if (int e = mhip_bbpgd_stage_finalize(op, init, gathered, c->world, stream)) return e;
    if (ev) MHIP_HIP(hipEventRecord(ev[6], s));
    return MHIP_SUCCESS;
  };

  // ONE exit: whatever happens between here and the end of the loop, the exchange numbers this solve may have used are
  // retired below (a rank that left them behind would meet its own stale inbox words in the next solve)
  unsigned enqueued = 0;
  auto run = [&]() -> int {
    if (int e = iteration(1, nullptr)) return e;
    unsigned last_todo = 0, iter_before = 0, chunk = 8, polls = 0;
    int done = 0;
    PollPlan plan;  // (only its rule for the last stretches: the residual is the same on every rank, so is the decision)
    for (;;) {
      if (int e = mhip_bbpgd_stage_poll(op, result, &done, stream)) return e;
      if (c->mbox.open || ipc)
        if (int e = mailbox_check(c, s)) return e;
      if (++polls == fault_at_poll)
        return fail(MHIP_ERR_RUNTIME, "rank %d: injected fault at poll %u (mhip_comm_inject_fault)", c->rank, polls);
      if (prof && last_todo) {
        unsigned eff = result->num_iters - iter_before + ((result->converged && result->num_iters < config->max_iters) ? 1u : 0u);
        if (eff > last_todo) eff = last_todo;
        for (unsigned k = 0; k < eff; k += kStride) {
          hipEvent_t* ev = &c->events[(size_t)kEv * (k / kStride)];
          float a = 0.f, p = 0.f, b = 0.f, w = 0.f, d = 0.f, r = 0.f;
          MHIP_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));
          MHIP_HIP(hipEventElapsedTime(&p, ev[1], ev[2]));
          MHIP_HIP(hipEventElapsedTime(&b, ev[2], ev[3]));
          MHIP_HIP(hipEventElapsedTime(&w, ev[3], ev[4]));
          MHIP_HIP(hipEventElapsedTime(&d, ev[4], ev[5]));
          MHIP_HIP(hipEventElapsedTime(&r, ev[5], ev[6]));
          profile->body_ms += a;
          profile->halo_post_ms += p;
          profile->constraint_ms += b + d;  // the wait for the halo is not part of the sweep's time
          profile->halo_wait_ms += w;
          profile->record_ms += r;
          profile->timed_iterations += 1;
        }
      }
      // (after a pause -- mhip_bbpgd_stage_poll has handled it -- the rest of the chunk did nothing: count what ran)
      if (!done && result->num_iters < enqueued) enqueued = result->num_iters;
      if (done || enqueued >= config->max_iters) break;
      if (enqueued >= 8 && !plan.shortened)  // the masks have settled: stream the active entries from a snapshot (convex.hip, OpView::aptr)
        if (int e = mhip_bbpgd_stage_snapshot_active(op, stream)) return e;
      iter_before = result->num_iters;
      // (stretches of 8, 16, 32, ... iterations up to poll_every, as in the fused driver: an easy solve is found converged
      // early, a long one is polled as rarely as the caller allows)
      const unsigned stretch = plan.stretch(chunk < poll_every ? chunk : poll_every, result->num_iters, result->residual,
                                            config->tol, C_global);
      if (chunk < poll_every) chunk *= 2;
      const unsigned todo = (config->max_iters - enqueued < stretch) ? config->max_iters - enqueued : stretch;
      for (unsigned k = 0; k < todo; ++k) {
        hipEvent_t* ev = (prof && k % kStride == 0) ? &c->events[(size_t)kEv * (k / kStride)] : nullptr;
        if (int e = iteration(0, ev)) {
          enqueued += k + 1;
          return e;
        }
      }
      enqueued += todo;
      last_todo = todo;
    }
    return MHIP_SUCCESS;
  };
  const int rc = run();
  // numbers of this solve's exchanges: seq_base (init) .. seq_base + iterations + 1.  A solve that ran to its end ran
  // the same count on every rank; one that failed retires everything it had enqueued (its peers' counts may differ then:
  // the next ghost plan re-agrees them)
  const unsigned used = (rc == MHIP_SUCCESS) ? result->num_iters : (enqueued > result->num_iters ? enqueued : result->num_iters);
  hi.seq_base = next_seq_base(seq_base, used);
  if (rc != MHIP_SUCCESS) {
    const std::string why = last_error_storage();   // (the clean-up below must not replace the message)
    (void)hipStreamSynchronize(s);
    if (c->in_flight) (void)mhip_comm_exchange_finish(c, stream);
    mhip_solve_result dummy{};
    (void)mhip_bbpgd_stage_end(op, &dummy, stream);
    (void)hipGetLastError();
    last_error_storage() = why;
    return rc;
  }
  return mhip_bbpgd_stage_end(op, result, stream);
}

int mhip_comm_set_exchange_timeout(mhip_comm_t c, double seconds) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  MHIP_REQUIRE(seconds >= 0.01 && seconds <= 3600.0, MHIP_ERR_INVALID_ARGUMENT, "timeout must be within [0.01, 3600] s");
  c->timeout_ticks = (unsigned long long)(seconds * 1e8);
  return MHIP_SUCCESS;
}

int mhip_comm_inject_fault(mhip_comm_t c, unsigned at_poll) {
  MHIP_REQUIRE(c != nullptr, MHIP_ERR_INVALID_ARGUMENT, "communicator is null");
  c->fault_at_poll = at_poll;
  return MHIP_SUCCESS;
}

}  // extern "C"
