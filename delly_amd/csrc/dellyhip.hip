// dellyhip.hip -- host side of the C-ABI declared in include/dellyhip.h.
// Thin marshalling only: every byte of DP work happens in the gfx950 kernels
// of split_kernel.hpp / split_main.hpp / msa_kernel.hpp.  There is no CPU
// fallback: without a usable device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dellyhip.h"
#include <thread>
#include <chrono>
#include "msa_kernel.hpp"
#include "split_main.hpp"
#include "split_pk.hpp"
#include "split_quad.hpp"
#include "split_sparse.hpp"
#include "ins_kernel.hpp"
#include "lr_kernel.hpp"
#include "lrmsa_kernel.hpp"
#include "lrins_kernel.hpp"
#include "lrwfa_kernel.hpp"
#include "classify_kernel.hpp"
#include "probes_kernel.hpp"
#include "edlib_kernel.hpp"
#include "comm.hpp"

static bool streams_run_concurrently(hipStream_t a, hipStream_t b, int* scratch2);   // (defined with the pipelined path)

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  g_err = buf;
  return code;
}

#define HIPCHK(x)                                          \
  do {                                                     \
    hipError_t _e = (x);                                   \
    if (_e != hipSuccess) return fail(DELLYHIP_E_RUNTIME, #x, _e); \
  } while (0)

// Device memory handed back by a DevBuf goes to a process-wide free list (per device) and is handed out again, never
// hipFree'd while the process lives (dellyhip_trim_memory does it on request).  Besides the usual reason -- hipMalloc and
// hipFree synchronise the device -- there is a measured one (tools/stream_matrix.sh, profiles/r03/README.md): device-to-host
// copies out of allocations made AFTER earlier allocations were freed ran at 9-34 GB/s instead of 46-58 GB/s, which cost
// every dellyhip_stream after the first of a process a third of its throughput (16.6 vs 23-24 M junctions/s).
struct DevPool {
  // dirty: handed back without a synchronisation -- work enqueued by the previous owner may still touch it (hipFree used
  // to wait implicitly).  The first reuse of a dirty block waits for the device once and clears the flag on every block.
  struct Block { void* p; size_t bytes; int device; bool dirty; uint64_t seq; };
  std::mutex mu;
  std::vector<Block> free_list;
  uint64_t seq = 0;   // blocks handed back so far (a device synchronisation cleans every block handed back before it started)
  // the device a block lives on: asked of the runtime, not assumed to be the calling thread's current device (a destructor
  // may run under another current device in a process that drives several GPUs)
  static int device_of(const void* p, bool host) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
    (void)hipGetLastError();
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)host;
    return dev;
  }
  static DevPool& get() { static DevPool* P = new DevPool(); return *P; }   // (never destroyed: the runtime may be gone at exit)
  static size_t round_up(size_t b) { return b <= (1u << 20) ? ((b + 255) & ~(size_t)255) : ((b + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1)); }
  void* take(size_t want, size_t* got, hipError_t* err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    want = round_up(std::max<size_t>(want, 1));
    {
      std::unique_lock<std::mutex> g(mu);
      size_t best = free_list.size();
      for (size_t i = 0; i < free_list.size(); ++i)
        if (free_list[i].device == dev && free_list[i].bytes >= want && free_list[i].bytes <= 2 * want + (1u << 16) &&
            (best == free_list.size() || free_list[i].bytes < free_list[best].bytes))
          best = i;
      if (best != free_list.size()) {
        Block b = free_list[best];
        free_list.erase(free_list.begin() + (long)best);
        if (!b.dirty) {
          *got = b.bytes;
          return b.p;
        }
        const uint64_t upto = seq;
        g.unlock();                            // (the wait for the device happens outside the lock: other threads go on allocating)
        (void)hipDeviceSynchronize();
        g.lock();
        for (auto& f : free_list)
          if (f.device == dev && f.seq <= upto) f.dirty = false;
        *got = b.bytes;
        return b.p;
      }
    }
    void* p = nullptr;
    *err = hipMalloc(&p, want);
    if (*err != hipSuccess) {   // out of memory with blocks parked here: give them back and try once more
      (void)hipGetLastError();
      if (trim(dev) == 0) return nullptr;
      *err = hipMalloc(&p, want);
      if (*err != hipSuccess) return nullptr;
    }
    *got = want;
    return p;
  }
  // Parked bytes per device are capped (half of the device's memory, DELLYHIP_POOL_LIMIT_MB overrides): beyond the cap the
  // oldest parked blocks go back to the runtime -- a process that keeps changing its batch shapes cannot hoard HBM.
  size_t limit_bytes(int dev) {
    if (const char* t = getenv("DELLYHIP_POOL_LIMIT_MB")) return (size_t)std::max(0ll, atoll(t)) << 20;
    static std::map<int, size_t> lim;   // (under mu)
    auto it = lim.find(dev);
    if (it != lim.end()) return it->second;
    size_t fr = 0, tot = 0;
    const size_t v = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? tot / 2 : ((size_t)64 << 30);
    lim[dev] = v;
    return v;
  }
  void give(void* p, size_t bytes) {
    const int dev = device_of(p, false);
    std::vector<Block> victims;
    {
      std::lock_guard<std::mutex> g(mu);
      free_list.push_back(Block{p, bytes, dev, true, ++seq});
      size_t parked = 0;
      for (auto& f : free_list)
        if (f.device == dev) parked += f.bytes;
      const size_t lim = limit_bytes(dev);
      for (size_t i = 0; parked > lim && i < free_list.size();)
        if (free_list[i].device == dev) {
          parked -= free_list[i].bytes;
          victims.push_back(free_list[i]);
          free_list.erase(free_list.begin() + (long)i);
        } else ++i;
    }
    for (auto& b : victims) (void)hipFree(b.p);   // (hipFree waits for the device: safe for blocks with work in flight)
  }
  size_t trim(int dev) {   // -> bytes returned to the runtime
    std::vector<Block> mine;
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size();)
        if (free_list[i].device == dev) { mine.push_back(free_list[i]); free_list.erase(free_list.begin() + (long)i); }
        else ++i;
    }
    size_t bytes = 0;
    for (auto& b : mine) { (void)hipFree(b.p); bytes += b.bytes; }
    return bytes;
  }
  // hipMalloc / hipFree shaped front end (dev_pool.hpp): the block sizes are remembered here
  std::map<void*, size_t> live;
  hipError_t raw_alloc(void** out, size_t bytes) {
    hipError_t e = hipSuccess;
    size_t got = 0;
    void* p = take(bytes, &got, &e);
    if (!p) { *out = nullptr; return e == hipSuccess ? hipErrorOutOfMemory : e; }
    std::lock_guard<std::mutex> g(mu);
    live[p] = got;
    *out = p;
    return hipSuccess;
  }
  void raw_free(void* p) {
    if (!p) return;
    size_t bytes = 0;
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = live.find(p);
      if (it == live.end()) return;
      bytes = it->second;
      live.erase(it);
    }
    give(p, bytes);
  }
};

// Owning device allocation: released by the destructor (every struct that holds one -- batches, job lists, the
// function-local buffers of the single-item wrappers -- gives its HBM back when it goes out of scope).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;
  bool owned = true;   // false: p points into another allocation (the staging block of a stream slot)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), cap_bytes(o.cap_bytes), owned(o.owned) { o.p = nullptr; o.n = 0; o.cap_bytes = 0; o.owned = true; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; cap_bytes = o.cap_bytes; owned = o.owned; o.p = nullptr; o.n = 0; o.cap_bytes = 0; o.owned = true; }
    return *this;
  }
  ~DevBuf() { release(); }
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return 0;
    hipError_t e = hipSuccess;
    p = static_cast<T*>(DevPool::get().take(count * sizeof(T), &cap_bytes, &e));
    if (!p) { n = 0; cap_bytes = 0; return fail(DELLYHIP_E_NOMEM, "hipMalloc", e); }
    return 0;
  }
  // keeps the allocation when it is already large enough
  int reserve(size_t count) { return (p && owned && n >= count) ? 0 : alloc(count); }
  // like reserve(), but a growing buffer gets 25 % head room (recycled batches of a stream: sizes wobble from batch to batch)
  int reserve_grow(size_t count) { return (p && owned && n >= count) ? 0 : alloc(count + count / 4 + 64); }
  void borrow(T* ptr, size_t count) {
    release();
    p = ptr;
    n = count;
    owned = false;
  }
  void release() {
    if (p && owned) DevPool::get().give(p, cap_bytes);
    p = nullptr;
    n = 0;
    cap_bytes = 0;
    owned = true;
  }
};

}  // namespace

namespace dh {
hipError_t dev_alloc(void** p, size_t bytes) { return DevPool::get().raw_alloc(p, bytes); }
void dev_free(void* p) { DevPool::get().raw_free(p); }
}  // namespace dh

// The resident chromosomes of one device.  Shared (ref-counted) by every context created with dellyhip_create_shared:
// the worker threads of the reference's ThreadPool (src/shortpe.h:175-201) and the slots of a dellyhip_stream all see ONE
// copy of the genome (3.1 GB), not one per context.
struct ChrTable {
  std::mutex mu;
  int device = 0;
  std::vector<uint8_t*> dev;
  std::vector<int64_t> len;
  uint64_t version = 1;
  ~ChrTable() {
    (void)hipSetDevice(device);
    for (auto p : dev)
      if (p) dh::dev_free(p);
  }
};

struct dellyhip_stream;

struct dellyhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true;   // (the slots of a dellyhip_stream run on the stream object's two compute streams)
  dellyhip_params params{};
  int n_cu = 0;
  // chromosome table: the shared object and this context's snapshot of it (refresh_chr)
  std::shared_ptr<ChrTable> chrs;
  uint64_t chr_seen = 0;
  std::vector<uint8_t*> chr_dev;   // device pointers
  std::vector<int64_t> chr_len;
  dellyhip_stream* host_streams[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the host-buffer entry points run through persistent one-slot streams (run_host_batch)
  DevBuf<const uint8_t*> d_chr_ptr;
  DevBuf<int64_t> d_chr_len;
  bool chr_dirty = true;
  // per-resident-block scratch of the split kernel
  DevBuf<uint32_t> scratch;
  uint64_t scratch_words = 0;
  int scratch_blocks = 0;
  DevBuf<int32_t> counters;  // work counters (one per K bin + MSA)
  hipEvent_t serial_ev = nullptr;  // end of the last batch_run: runs of one context share its scratch area, so a run on
                                   // another stream first waits for it (overlap batches with one context per stream)
  bool serial_valid = false;
  hipStream_t serial_stream = nullptr;   // the stream serial_ev was recorded on
  int msa_tmax = dh::TMAXC;  // column types per MSA node served by the score table (env DELLYHIP_MSA_TMAX)
  int wfa_lds_seed = 1;      // msaWfa's diagonal seeding in its own kernel with the k-mer table in LDS (wfa_seed_kernel; env DELLYHIP_WFA_LDS_SEED=0: inside wfa_pairs_kernel, tables in HBM)
  int myers_band = 1;        // banded bit-vector distances, several pairs per wavefront (myers_band.hpp; env DELLYHIP_MYERS_BAND=0: the full passes)
  int lrc_waves = 2;         // resident wavefronts per SIMD of the long-read consensus kernels (lrmsa_kernel / lrwfa_kernel: one junction per wavefront,
                             // 145 / 171 VGPRs = 3 / 2 per SIMD by registers; env DELLYHIP_LRC_WAVES; round 5 launched one per SIMD)
  int lr_waves = 8;          // resident wavefronts of the strip kernel per CU when the sparse passes are on (env DELLYHIP_LR_WAVES)
  int lr_team_serial = 0;    // env DELLYHIP_LR_TEAMS_SERIAL (A/B)
  int lr_teams = 64;         // teams of lr_dense_team_kernel at most (env DELLYHIP_LR_TEAMS; 0: the dense strips stay on lr_kernel's wavefronts)
  hipStream_t lr_aux = nullptr;            // the teams' stream (beside the stream of the batch); lr_aux_all: every candidate created for it
  std::vector<hipStream_t> lr_aux_all;
  int sps_waves = 16;        // wavefronts of split_sparse_kernel per CU (env DELLYHIP_SPS_WAVES; 16 = what LDS and registers allow)
  int sr_sparse = 1;         // short-read shapes through split_sparse_kernel first (env DELLYHIP_SR_SPARSE=0: dense kernels only)
  int sr_wide = 1;           // ... and the shapes beyond its byte tile through split_sparse_wide_kernel (env DELLYHIP_SR_WIDE=0: dense kernels, A/B)
  DevBuf<uint32_t> spw_scratch;   // its per-block tables (allocated at first use)
  int spw_blocks = 0;
  int sparse_cost = 160;     // predicted deficit up to which the sparse passes go on (env DELLYHIP_SPARSE_COST; tuning)
  int use_sparse = 1;        // sparse (furthest-reaching) longNeedle in the strip kernel (env DELLYHIP_SPARSE=0: dense passes only)
  int use_quad = 1;          // four junctions per wavefront where they fit (env DELLYHIP_QUAD=0: packed pairs only)
  int msa_only = 0;          // env DELLYHIP_MSA_ONLY=1 (profiling builds): msa() batches stop after the MSA kernels
  int msa_waves = 16;        // resident wavefronts of msa_kernel per CU (env DELLYHIP_MSA_WAVES; 128 VGPRs and 9.7 KB of LDS allow 16)
  int msa_team = 0;          // wavefronts per junction of the MSA kernel: 0 = by batch size (env DELLYHIP_MSA_TEAM = 1 | 2 | 4 forces it)
  int msa_pair = 1;          // two merges of a junction per Gotoh pass where they fit (env DELLYHIP_MSA_PAIR=0: one merge per pass, A/B)
  int quad_mix = 0;          // env DELLYHIP_QUAD_MIX=1: top whole quad rounds up with pair items
};

struct SmallInv { int32_t j, full_len, offset, take; };   // long-read loop, small inversions (src/assemble.h:840-853)

struct dellyhip_batch {
  bool ever_run = false;
  bool lr_failed = false;   // a long-read team gave up in the last run: every sync / fetch / gather until the next run fails
  int probe_mode = 0;   // _generateProbes flavour: no early length test (src/split.h:647), bit 1 of params.reserved on the device
  int32_t n = 0;
  uint64_t n_seq = 0;
  int with_msa = 0;
  int want_alignment = 0;
  std::vector<dellyhip_junction> h_junc;
  std::vector<int32_t> h_cons_len;   // U path: known on the host
  DevBuf<dellyhip_junction> junc;
  DevBuf<uint8_t> seq_blob;
  DevBuf<uint64_t> seq_off;
  DevBuf<uint64_t> cons_off;
  DevBuf<int32_t> cons_len;
  DevBuf<dellyhip_result> res;
  DevBuf<uint8_t> out_blob;
  uint64_t out_stride = 0;
  int32_t out_cons_cap = dh::OUT_CONS_CAP, out_allele_cap = dh::OUT_ALLELE_CAP, out_aln_cap = dh::OUT_ALN_CAP;
  DevBuf<int32_t> work;              // K-binned pair lists, concatenated
  std::vector<int32_t> bin_first, bin_count;  // per K = 1..KMAX
  int ins_first = 0, ins_count = 0;  // svt 4 junctions (insertion kernel): work[ins_first .. +ins_count)
  bool sps_all = false;              // every junction of the dense bins is in the sparse list too
  bool zero_res_pending = false;     // the records are to be zeroed before the split kernels of this run (a re-run without an MSA stage)
  bool sps_identity = false;         // the sparse list is 0, 1, 2, ...: the kernel is launched without it (one dependent load less per junction)
  // msa() batches: the sparse kernel runs on every non-insertion junction straight behind the MSA kernels (it reads the
  // consensus lengths on the device) while the host routes the batch from the downloaded lengths
  DevBuf<int32_t> early_list;
  DevBuf<int32_t> msa_order;         // msa() batches whose junctions differ in cost: the order the MSA kernels take them in (most expensive first)
  bool msa_ordered = false;
  int early_count = 0;
  bool early_done = false;           // this run: split_sparse_kernel has been launched already
  int lazy = 0;                      // stream slot: dense routing of the sparse kernel's leftovers happens at collect time
  bool lazy_pending = false;         // ... and has not happened yet for the current run
  int32_t* pin_cons_len = nullptr;   // stream slot: pinned destination of the consensus lengths (msa() batches)
  int32_t* own_pin_len = nullptr;    // resident batches: the same, owned by the batch (an asynchronous download into PAGEABLE memory goes
                                     // through the runtime's staging path and was measured to add 0.5 ms per step in a process that had
                                     // used several streams before -- bench.py's u_full row after the headline leg, round 5)
  size_t own_pin_bytes = 0, own_pin_count = 0;
  int sps_first = 0, sps_count = 0;  // junctions split_sparse_kernel tries first (they also sit in a dense bin)
  int spw_first = 0, spw_count = 0;  // ... and those beyond its byte tile that split_sparse_wide_kernel tries first (they sit in a dense bin too)
  bool sr_wide = false;
  int sr_sparse = 1;
  // four-junctions-per-wavefront bins (|consensus| <= 159): work[qbin_first[Kq] .. ) holds 4 indices per item
  std::vector<int32_t> qbin_first, qbin_count, qbin_pairs;   // per KQ: offset, quad items, pair items behind them
  int use_quad = 1, quad_mix = 0;
  int n_simd = 1024;
  // long-read shapes (|consensus| > 319 or |svRefStr| > 2048): strip kernel, per-block workspace
  std::vector<int32_t> h_win_len;    // |svRefStr| per junction, computed on the host (U path)
  int lr_first = 0, lr_count = 0, lr_blocks = 0;
  dh::LrArgs lr{};
  DevBuf<uint8_t> lr_ws;
  int lr_teams = 0;                  // teams of lr_dense_team_kernel for this batch (0: none)
  DevBuf<int32_t> lr_team_state;     // dh::LRT_* counters + the list of junctions handed to the teams
  hipEvent_t lr_fork = nullptr, lr_join = nullptr;
  bool lr_aux_used = false;          // a kernel of this batch was launched on the context's auxiliary stream
  hipEvent_t lri_fork = nullptr, lri_join = nullptr;   // lr_ins_kernel beside the dense / insertion kernels (run_split_dense)
  // long-read insertions (svt 4 beyond the short-read shapes)
  int lri_first = 0, lri_count = 0, lri_blocks = 0;
  dh::LrInsArgs lri{};
  DevBuf<uint8_t> lri_ws;
  // long-read MSA of insertions (with_msa == 2, svt 4: msaWfa)
  DevBuf<int32_t> wfa_list;
  DevBuf<uint8_t> wfa_ws;
  // the pairwise stage of msaWfa as its own kernel (wfa_pairs_kernel): one (junction, read pair) per wavefront
  DevBuf<int32_t> wfa_pair_first, wfa_edit, wfa_seeds;
  int wfa_max_rows = 0;
  DevBuf<uint8_t> wfa_pair_ws;
  DevBuf<uint32_t> wfa_next;
  dh::WfaPairArgs wfa_pairs{};
  int wfa_items = 0, wfa_pair_grid = 1;
  dh::LrWfaArgs wfa{};
  int wfa_count = 0, wfa_blocks = 0;
  DevBuf<SmallInv> small_inv;
  int small_inv_n = 0;
  // long-read MSA (with_msa == 2: msaEdlib)
  DevBuf<int32_t> lm_edit, lm_pair_first;
  DevBuf<int8_t> lm_hbuf;            // strip passes of read pairs beyond MYERS_ROWS (2 x lm_hbuf_half bytes per pair wavefront)
  uint64_t lm_hbuf_half = 0;
  int lm_pair_grid = 1;
  DevBuf<uint8_t> lm_ws;
  dh::LrMsaArgs lm{};
  int lm_items = 0, lm_blocks = 0;
  int lm_maxlen = 1;         // longest read of a long-read consensus batch
  // dellyhip_batch_fetch: device-side compaction
  DevBuf<uint64_t> blob_off;
  DevBuf<uint8_t> blob_compact;
  // dellyhip_batch_fetch_begin / _end: the same compaction queued behind the batch's kernels, its output written by a kernel into
  // the caller's pinned host memory
  hipStream_t run_stream = nullptr;  // the stream of the last dellyhip_batch_run
  hipStream_t fetch_stream = nullptr;
  hipEvent_t fetch_ev = nullptr;
  bool fetch_pending = false;
  uint64_t* fetch_status = nullptr;  // pinned: [0] = blob bytes, [1] = flags (1: the caller's blob area is too small, 2: a long-read team gave up)
  size_t fetch_status_bytes = 0;
  // direct (single longNeedle) mode
  DevBuf<uint8_t> ref_blob;
  DevBuf<uint64_t> ref_off;
  DevBuf<int32_t> ref_len;
  // MSA stage
  DevBuf<uint8_t> msa_ws;
  DevBuf<uint8_t> msa_big_ws;        // msa_big instance: junctions beyond the standard instance's shapes
  dh::MsaPlan msa_plan;
  int msa_team = 1, msa_grid = 1;     // wavefronts per junction of the score-table MSA kernel, its blocks
  uint64_t msa_stride = 0;            // workspace bytes per block
  int msa_big_grid = 0;
  // timing
  std::vector<hipEvent_t> ev;        // 4 events per launch since the last kernel_ms()
  std::vector<hipEvent_t> ev_free;   // recycled events (hipEventCreate costs ~10 us each: not inside a pipelined loop)
  hipEvent_t last = nullptr;
  hipEvent_t mid = nullptr;          // recorded between the DP kernels and the post kernel
  hipEvent_t len_ev = nullptr;       // msa() batches: the consensus lengths have arrived on the host
  double ms_split = 0, ms_msa = 0, ms_dp = 0, ms_dp_last = 0;
  int launches = 0;
  bool pending = false;
};

namespace {

// this context's view of the shared chromosome table
void refresh_chr(dellyhip_ctx* c) {
  std::lock_guard<std::mutex> g(c->chrs->mu);
  if (c->chr_seen == c->chrs->version) return;
  c->chr_dev = c->chrs->dev;
  c->chr_len = c->chrs->len;
  c->chr_seen = c->chrs->version;
  c->chr_dirty = true;
}

int ensure_chr_table(dellyhip_ctx* c) {
  refresh_chr(c);
  if (!c->chr_dirty) return 0;
  size_t n = c->chr_dev.size();
  if (n == 0) return fail(DELLYHIP_E_ARG, "no chromosome uploaded (dellyhip_set_chromosome)");
  int rc;
  HIPCHK(hipDeviceSynchronize());   // (kernels in flight may still read the old table)
  if ((rc = c->d_chr_ptr.alloc(n))) return rc;
  if ((rc = c->d_chr_len.alloc(n))) return rc;
  HIPCHK(hipMemcpy(c->d_chr_ptr.p, c->chr_dev.data(), n * sizeof(uint8_t*), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_chr_len.p, c->chr_len.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));
  c->chr_dirty = false;
  return 0;
}

int ensure_scratch(dellyhip_ctx* c) {
  if (c->scratch.p) return 0;
  const uint64_t nblk = (dh::NMAX + 63 + 15) / 16 + 1;
  c->scratch_words = nblk * 2 * dh::KMAX * dh::WAVE;  // packed pair stack: 2 dwords per 16 steps per slot
  c->scratch_words = std::max<uint64_t>(c->scratch_words, dh::sps_scratch_bytes() / 4 + 64);   // tables of split_sparse_kernel
  c->scratch_blocks = c->n_cu * 20;  // 5 waves per SIMD resident
  int rc = c->scratch.alloc((size_t)c->scratch_words * c->scratch_blocks);
  if (rc) return rc;
  return c->counters.alloc(32);
}

// three launches per K bin: packed DP over pairs, 32-bit kernel for deferred junctions,
// post-processing (tracebacks, split detection) one junction per wavefront
template <int K>
void launch_split(dh::SplitArgs a, int pairs, int max_blocks, int32_t* counters, hipStream_t s, hipEvent_t mid) {
  // equal number of work items per block (static striding): a grid of `max_blocks` with a
  // remainder would leave most of the chip idle during the last partial round
  auto balanced = [&](int items) {
    int rounds = (items + max_blocks - 1) / max_blocks;
    return (items + rounds - 1) / rounds;
  };
  a.n_work = pairs;
  a.work_counter = counters + K;
  hipLaunchKernelGGL(dh::split_pair_kernel<K>, dim3(balanced(pairs)), dim3(dh::WAVE), 0, s, a);
  a.n_work = 2 * pairs;  // same counter: number of deferred pairs (0 -> the kernel returns at once)
  hipLaunchKernelGGL(dh::split_align_kernel<K>, dim3(std::min({2 * pairs, 1024, max_blocks})), dim3(dh::WAVE), 0, s, a);   // (scratch holds max_blocks blocks)
  if (mid) (void)hipEventRecord(mid, s);  // (with several K bins: the last bin's DP end)
  hipLaunchKernelGGL(dh::split_post_kernel<K>, dim3(balanced(2 * pairs)), dim3(dh::WAVE), 0, s, a);
}

// quad bins: packed DP over four junctions per wavefront (rows per lane KQ in a 32-lane half), then the
// 32-bit kernel for deferred junctions and the post kernel with the matching 64-lane rows-per-lane KP
template <int KQ, int KP>
void launch_quad(dh::SplitArgs a, int n_quads, int n_pairs, int max_blocks, int32_t* counters, hipStream_t s, hipEvent_t mid) {
  auto balanced = [&](int n) {
    int rounds = (n + max_blocks - 1) / max_blocks;
    return (n + rounds - 1) / rounds;
  };
  const int items = n_quads + n_pairs, seats = 4 * n_quads + 2 * n_pairs;
  a.n_work = items;
  a.work_counter = counters + 5 + KQ;   // (slots 6..10; the deferred count lives 16 further)
  hipLaunchKernelGGL((dh::split_quad_kernel<KQ, KP>), dim3(balanced(items)), dim3(dh::WAVE), 0, s, a, n_quads);
  a.n_work = seats;   // the list is a flat array of junction indices (-1 = empty seat) for the next two kernels
  hipLaunchKernelGGL(dh::split_align_kernel<KP>, dim3(std::min({seats, 1024, max_blocks})), dim3(dh::WAVE), 0, s, a);
  if (mid) (void)hipEventRecord(mid, s);
  hipLaunchKernelGGL(dh::split_post_kernel<KP>, dim3(balanced(seats)), dim3(dh::WAVE), 0, s, a);
}

dh::SplitArgs make_split_args(dellyhip_ctx* c, dellyhip_batch* b, bool direct);
int run_split_dense(dellyhip_ctx* c, dellyhip_batch* b, hipStream_t s, bool direct, dh::SplitArgs a, bool mid_done);

// split_sparse_wide_kernel over a list, behind split_sparse_kernel on the same stream (counted: see the kernel)
int launch_sparse_wide(dellyhip_ctx* c, dh::SplitArgs a, const int32_t* list, int count, int counted, hipStream_t s) {
  if (count <= 0) return 0;
  // per-block tables (~428 KB each): sized from the list, not from the chip -- the kernel normally sees a handful of junctions and
  // every slot of a dellyhip_stream is a context of its own (512 blocks were 224 MB per context, 1.3 GB per depth-6
  // stream).  Grows (rarely) behind a stream synchronisation; dellyhip_trim_memory releases it.
  const int want = std::min(std::max(64, std::min(512, c->n_cu * 2)), std::max(32, count));
  if (!c->spw_scratch.p || c->spw_blocks < want) {
    if (c->spw_scratch.p) (void)hipDeviceSynchronize();   // (an earlier launch, on whatever stream, may still be using the smaller block)
    int blocks = 32;
    while (blocks < want) blocks *= 2;
    int rc = c->spw_scratch.alloc((size_t)(dh::spw_scratch_bytes() / 4) * blocks);
    if (rc) { c->spw_blocks = 0; return rc; }
    c->spw_blocks = blocks;
  }
  a.work_list = list;
  a.n_work = count;
  a.work_counter = nullptr;
  hipLaunchKernelGGL(dh::split_sparse_wide_kernel, dim3(std::min(count, c->spw_blocks)), dim3(dh::WAVE), 0, s, a, c->spw_scratch.p,
                     (uint64_t)(dh::spw_scratch_bytes() / 4), counted);
  HIPCHK(hipGetLastError());
  return 0;
}

// msa() batches: split_sparse_kernel on every non-insertion junction, enqueued straight behind the MSA kernels
int launch_early_sparse(dellyhip_ctx* c, dellyhip_batch* b, hipStream_t s) {
  int rc;
  if ((rc = ensure_chr_table(c))) return rc;
  dh::SplitArgs a = make_split_args(c, b, false);
  a.work_list = b->early_list.p;
  a.n_work = b->early_count;
  a.work_counter = c->counters.p + 30;
  a.sps_left = c->counters.p + 31;
  hipLaunchKernelGGL(dh::split_sparse_kernel, dim3(std::min({b->early_count, c->scratch_blocks, c->n_cu * c->sps_waves})), dim3(dh::WAVE), 0, s, a);
  HIPCHK(hipGetLastError());
  if (b->mid) HIPCHK(hipEventRecord(b->mid, s));
  // (the consensus lengths are still on the device: the wide kernel looks at every junction the first one left and takes
  //  those whose shape was beyond it)
  if (b->sr_wide && (rc = launch_sparse_wide(c, a, b->early_list.p, b->early_count, 1, s))) return rc;
  b->early_done = true;
  return 0;
}

// Two regions zeroed by ONE kernel launch.  hipMemsetAsync is a fill kernel behind a slower submission path: in the kernel trace of
// the headline arrangement each of the two memsets of a step (the batch's records on a re-run, the work counters) sat behind a
// gap of 16 / 6 us, 31 us of a stream's 330 us cycle with the fills themselves -- a kernel of this library follows its
// predecessor without a gap.
__global__ __launch_bounds__(256) void zero2_kernel(uint4* a, size_t na16, uint32_t* b, int nb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t k = i; k < na16; k += (size_t)gridDim.x * blockDim.x) a[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < (size_t)nb) b[i] = 0u;
}
static int launch_zero2(hipStream_t s, void* a, size_t a_bytes, void* b, int b_words) {
  if (a_bytes % 16 || (reinterpret_cast<uintptr_t>(a) & 15)) {   // (never the case for the record array: 144-byte records in a hipMalloc block)
    HIPCHK(hipMemsetAsync(a, 0, a_bytes, s));
    a_bytes = 0;
  }
  const size_t na16 = a_bytes / 16;
  const int blocks = (int)std::min<size_t>(std::max<size_t>((std::max<size_t>(na16, (size_t)b_words) + 255) / 256, 1), 2048);
  hipLaunchKernelGGL(zero2_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(a), na16, reinterpret_cast<uint32_t*>(b), b_words);
  HIPCHK(hipGetLastError());
  return 0;
}

// Launches the split-alignment kernels for every K bin of the batch.
int run_split(dellyhip_ctx* c, dellyhip_batch* b, hipStream_t s, bool direct) {
  int rc;
  if (!direct && (rc = ensure_chr_table(c))) return rc;
  if ((rc = ensure_scratch(c))) return rc;
  const bool early = b->early_done;   // (slots 30 / 31 of the counters belong to the kernel that already ran)
  b->early_done = false;
  {   // the work counters, and -- on a re-run without an MSA stage -- the batch's records (dellyhip_batch_run)
    const bool zr = b->zero_res_pending;
    b->zero_res_pending = false;
    if ((rc = launch_zero2(s, zr ? (void*)b->res.p : nullptr, zr ? (size_t)b->n * sizeof(dellyhip_result) : 0, c->counters.p, early ? 30 : 32))) return rc;
  }
  dh::SplitArgs a = make_split_args(c, b, direct);
  bool mid_done = false;
  if (early) {   // every junction of the dense bins was offered to the sparse kernel
    a.sps_left = c->counters.p + 31;
    mid_done = true;
  } else if (b->sps_count > 0 && !direct) {   // sparse longNeedle first: the dense kernels below skip what it finishes
    a.work_list = b->sps_identity ? nullptr : b->work.p + b->sps_first;
    a.n_work = b->sps_count;
    a.work_counter = c->counters.p + 30;
    a.sps_left = c->counters.p + 31;
    hipLaunchKernelGGL(dh::split_sparse_kernel, dim3(std::min({b->sps_count, c->scratch_blocks, c->n_cu * c->sps_waves})), dim3(dh::WAVE), 0, s, a);
    HIPCHK(hipGetLastError());
    if (b->spw_count > 0 && (rc = launch_sparse_wide(c, a, b->work.p + b->spw_first, b->spw_count, 0, s))) return rc;
    if (!b->sps_all) a.sps_left = nullptr;   // some junction of the dense bins was never offered to a sparse kernel
    else if (b->mid) { HIPCHK(hipEventRecord(b->mid, s)); mid_done = true; }   // dp_kernel_ms = the sparse kernel (the dominant one)
  } else if (b->spw_count > 0 && !direct) {   // (no junction for the byte-tile kernel at all)
    a.sps_left = c->counters.p + 31;
    if ((rc = launch_sparse_wide(c, a, b->work.p + b->spw_first, b->spw_count, 0, s))) return rc;
    a.sps_left = nullptr;
  }
  return run_split_dense(c, b, s, direct, a, mid_done);
}

// the stream lr_dense_team_kernel runs on, beside the stream of the batch; the fork / join events of the batch.  HIP streams
// share a few hardware queues and two streams on one queue run in order (the teams would start when lr_kernel ends: correct,
// but nothing gained), so candidates are probed against the context's own stream as dellyhip_compute_streams probes its pair.
int ensure_lr_aux(dellyhip_ctx* c, dellyhip_batch* b) {
  if (!b->lr_fork) HIPCHK(hipEventCreateWithFlags(&b->lr_fork, hipEventDisableTiming));
  if (!b->lr_join) HIPCHK(hipEventCreateWithFlags(&b->lr_join, hipEventDisableTiming));
  if (c->lr_aux) return 0;
  // ONE auxiliary stream per device and process, found by the first context that needs one: the slots of a dellyhip_stream are
  // contexts of their own, and every stream a process creates competes for the few hardware queues (CHANGELOG.md 1b)
  static std::mutex aux_mu;
  static std::map<int, hipStream_t> aux_of_device;
  std::lock_guard<std::mutex> aux_guard(aux_mu);
  {
    auto it = aux_of_device.find(c->device);
    if (it != aux_of_device.end()) { c->lr_aux = it->second; return 0; }
  }
  int* probe = nullptr;
  const bool can_probe = !getenv("DELLYHIP_STREAM_NO_PROBE") && dh::dev_alloc((void**)&probe, 2 * sizeof(int)) == hipSuccess;
  hipStream_t pick = nullptr;
  for (int t = 0; t < 4 && !pick; ++t) {
    hipStream_t cand = nullptr;
    if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
    c->lr_aux_all.push_back(cand);
    if (!can_probe || streams_run_concurrently(c->stream, cand, probe)) pick = cand;
  }
  if (probe) dh::dev_free(probe);
  if (!pick && !c->lr_aux_all.empty()) pick = c->lr_aux_all[0];
  if (!pick) return fail(DELLYHIP_E_RUNTIME, "no stream for the long-read teams");
  c->lr_aux = pick;
  aux_of_device[c->device] = pick;
  return 0;
}

dh::SplitArgs make_split_args(dellyhip_ctx* c, dellyhip_batch* b, bool direct) {
  dh::SplitArgs a{};
  a.junc = b->junc.p;
  a.cons_off = b->cons_off.p;
  a.cons_len = b->cons_len.p;
  a.cons_base = b->with_msa ? b->out_blob.p : b->seq_blob.p;
  a.chr_seq = c->d_chr_ptr.p;
  a.chr_len = c->d_chr_len.p;
  a.n_chr = (int)c->chr_dev.size();
  a.p = c->params;
  if (b->probe_mode) a.p.reserved |= 2;
  a.res = b->res.p;
  a.out_blob = b->out_blob.p;
  a.out_stride = b->out_stride;
  a.scratch = c->scratch.p;
  a.scratch_words = c->scratch_words;
  a.want_alignment = b->want_alignment;
  a.out_cons_cap = b->out_cons_cap;
  a.out_allele_cap = b->out_allele_cap;
  a.pair_mode = 1;
  if (direct) {
    a.ref_base = b->ref_blob.p;
    a.ref_off = b->ref_off.p;
    a.ref_len = b->ref_len.p;
  }
  return a;
}

static bool env_on(const char* name) {
  const char* t = getenv(name);
  return t && atoi(t) != 0;
}

int run_split_dense(dellyhip_ctx* c, dellyhip_batch* b, hipStream_t s, bool direct, dh::SplitArgs a, bool mid_done) {
  // Long insertions (svt 4 beyond the short-read shapes: a consensus of 20 reads over a 60-120 bp insertion is 320-420 bp) are
  // few per batch and take ~0.8 ms each on one wavefront: a latency chain, not a load.  Round 5: lr_ins_kernel starts FIRST, on
  // the context's auxiliary stream, beside the dense / insertion kernels of the other bins (disjoint junctions, its own
  // workspace; it must follow the sparse kernel, which rewrites the default record of every junction it is offered) and the
  // batch's stream joins it at the end: the all-SV-types row of bench.py 2.1 -> 1.4 ms of split stage per 10 000 junctions
  // (DELLYHIP_LRI_SERIAL=1: on the batch's own stream, after everything else, as before).
  bool lri_side = false;
  if (b->lri_count > 0 && !direct && !env_on("DELLYHIP_LRI_SERIAL") && ensure_lr_aux(c, b) == 0 && c->lr_aux != s) {
    if (!b->lri_fork) HIPCHK(hipEventCreateWithFlags(&b->lri_fork, hipEventDisableTiming));
    if (!b->lri_join) HIPCHK(hipEventCreateWithFlags(&b->lri_join, hipEventDisableTiming));
    HIPCHK(hipEventRecord(b->lri_fork, s));
    HIPCHK(hipStreamWaitEvent(c->lr_aux, b->lri_fork, 0));
    dh::SplitArgs ai = a;
    ai.sps_left = nullptr;
    ai.work_list = b->work.p + b->lri_first;
    ai.n_work = b->lri_count;
    const int rounds = (b->lri_count + b->lri_blocks - 1) / b->lri_blocks;
    const int grid = (b->lri_count + rounds - 1) / rounds;
    dh::LrInsArgs li = b->lri;
    li.realign = ((c->params.reserved & 1) && b->with_msa != 2 && !b->ref_blob.p) ? 1 : 0;   // src/assemble.h:859: the long-read loop passes realign = false for insertions (direct pairs: splitAlign alone)
    hipLaunchKernelGGL(dh::lr_ins_kernel, dim3(grid), dim3(dh::WAVE), 0, c->lr_aux, ai, li);
    hipError_t e1 = hipGetLastError();
    if (e1 == hipSuccess) e1 = hipEventRecord(b->lri_join, c->lr_aux);
    if (e1 != hipSuccess) {
      (void)hipStreamSynchronize(c->lr_aux);
      return fail(DELLYHIP_E_RUNTIME, "lr_ins_kernel on the auxiliary stream", e1);
    }
    b->lr_aux_used = true;
    lri_side = true;
  }
  struct JoinSide {   // every return path below joins the side launch
    dellyhip_ctx* c; dellyhip_batch* b; hipStream_t s; bool on;
    ~JoinSide() { if (on && hipStreamWaitEvent(s, b->lri_join, 0) != hipSuccess) (void)hipStreamSynchronize(c->lr_aux); }
  } join_side{c, b, s, lri_side};
  bool any_bin = false;
  for (int K = 1; K <= dh::KMAX; ++K) {
    int cnt = b->bin_count[K];
    if (cnt == 0) continue;
    any_bin = true;
    a.work_list = b->work.p + 2 * b->bin_first[K];
    switch (K) {
      case 1: launch_split<1>(a, cnt, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 2: launch_split<2>(a, cnt, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 3: launch_split<3>(a, cnt, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 4: launch_split<4>(a, cnt, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      default: launch_split<5>(a, cnt, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
    }
    HIPCHK(hipGetLastError());
  }
  for (int KQ = 1; KQ <= 5 && !b->qbin_count.empty(); ++KQ) {
    const int cnt = b->qbin_count[KQ], np = b->qbin_pairs[KQ];
    if (cnt + np == 0) continue;
    any_bin = true;
    a.work_list = b->work.p + b->qbin_first[KQ];
    switch (KQ) {
      case 1: launch_quad<1, 1>(a, cnt, np, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 2: launch_quad<2, 1>(a, cnt, np, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 3: launch_quad<3, 2>(a, cnt, np, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      case 4: launch_quad<4, 2>(a, cnt, np, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
      default: launch_quad<5, 3>(a, cnt, np, c->scratch_blocks, c->counters.p, s, mid_done ? nullptr : b->mid); break;
    }
    HIPCHK(hipGetLastError());
  }
  if (!any_bin && b->mid && !mid_done) HIPCHK(hipEventRecord(b->mid, s));  // keeps the per-launch event quartet complete
  a.sps_left = nullptr;
  if (b->ins_count > 0) {   // (direct mode: dellyhip_split_align)
    a.work_list = b->work.p + b->ins_first;
    a.n_work = b->ins_count;
    const int rounds = (b->ins_count + c->scratch_blocks - 1) / c->scratch_blocks;
    const int grid = (b->ins_count + rounds - 1) / rounds;
    hipLaunchKernelGGL(dh::ins_kernel, dim3(grid), dim3(dh::WAVE), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  if (b->lr_count > 0) {   // (direct: dellyhip_long_needle beyond the short-read shapes)
    a.work_list = b->work.p + b->lr_first;
    a.n_work = b->lr_count;
    a.work_counter = c->counters.p + 28;
    const int grid = std::min(b->lr_count, b->lr_blocks);
    dh::LrArgs lr = b->lr;
    lr.realign = ((c->params.reserved & 1) && !b->ref_blob.p) ? 1 : 0;   // (direct pairs: longNeedle alone, no orientation test)
    lr.lr_grid = grid;
    hipStream_t aux = nullptr;
    if (b->lr_teams > 0 && lr.team_state && ensure_lr_aux(c, b) == 0) aux = (s == c->lr_aux) ? c->stream : c->lr_aux;
    const bool serial = aux && c->lr_team_serial;   // (A/B: the teams after lr_kernel on the batch's own stream)
    if (aux) {
      HIPCHK(hipMemsetAsync(lr.team_state, 0, dh::LRT_LIST * sizeof(int32_t), s));
      HIPCHK(hipMemsetAsync(lr.team_state + dh::LRT_LIST, 0xff, (size_t)lr.team_cap * sizeof(int32_t), s));
#ifdef DH_LR_TEAM_DEBUG
      HIPCHK(hipMemsetAsync(lr.team_state + dh::LRT_LIST + lr.team_cap, 0, (size_t)dh::LRT_DBG_INTS * sizeof(int32_t), s));
#endif
      HIPCHK(hipEventRecord(b->lr_fork, s));
    } else {
      lr.team_state = nullptr;
    }
    hipLaunchKernelGGL(dh::lr_kernel, dim3(grid), dim3(dh::WAVE), 0, s, a, lr);
    HIPCHK(hipGetLastError());
    if (serial) {
      hipLaunchKernelGGL(dh::lr_dense_team_kernel, dim3(b->lr_teams), dim3(dh::WAVE * dh::LR_TEAM_W), 0, s, a, lr);
      HIPCHK(hipGetLastError());
    } else if (aux) {   // the teams run beside lr_kernel; the batch's stream goes on when both are through
      HIPCHK(hipStreamWaitEvent(aux, b->lr_fork, 0));
      hipLaunchKernelGGL(dh::lr_dense_team_kernel, dim3(b->lr_teams), dim3(dh::WAVE * dh::LR_TEAM_W), 0, aux, a, lr);
      // from here on a failure must not leave the teams running behind the caller's back (they poll the batch's workspace and
      // team state for up to ten minutes; batch_free / the next run only know the batch's own stream): join before returning
      hipError_t e1 = hipGetLastError();
      if (e1 == hipSuccess) e1 = hipEventRecord(b->lr_join, aux);
      if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, b->lr_join, 0);
      if (e1 != hipSuccess) {
        (void)hipStreamSynchronize(aux);
        return fail(DELLYHIP_E_RUNTIME, "long-read teams: launch / join", e1);
      }
      b->lr_aux_used = true;
    }
  }
  if (b->lri_count > 0 && !lri_side) {   // (direct: dellyhip_split_align beyond the short-read shapes)
    a.work_list = b->work.p + b->lri_first;
    a.n_work = b->lri_count;
    const int rounds = (b->lri_count + b->lri_blocks - 1) / b->lri_blocks;
    const int grid = (b->lri_count + rounds - 1) / rounds;
    dh::LrInsArgs li = b->lri;
    li.realign = ((c->params.reserved & 1) && b->with_msa != 2 && !b->ref_blob.p) ? 1 : 0;   // src/assemble.h:859: the long-read loop passes realign = false for insertions (direct pairs: splitAlign alone)
    hipLaunchKernelGGL(dh::lr_ins_kernel, dim3(grid), dim3(dh::WAVE), 0, s, a, li);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// |svRefStr| of a junction: the integer arithmetic of _initBreakpoint (src/tags.h:151-172) and of
// the concatenations in _getSVRef (src/split.h:70-163), mirrored from window_segments() so that
// the host can route a junction to the short-read or the long-read kernel and size workspaces
int host_window_len(const dellyhip_params& P, const dellyhip_junction& J, int m, const std::vector<int64_t>& chr_len) {
  auto clampz = [](long v) { return (int)std::max<long>(0, v); };
  // alignConsensus returns false before it builds a window when the consensus is shorter than both flanks + the insertion
  // (src/split.h:647).  For an insertion with |consensus| < insLen the window formula below wraps (size_t arithmetic, :651) to the
  // whole chromosome: such a junction -- consensus beyond the short-read kernels, "window" beyond the long-read ones -- used to be
  // routed to the short-read insertion kernel and came back as E_LIMIT where the reference says false (found by
  // tests/test_gpu_band.py, round 6; rounds 2-5 had it).  No window: the long-read insertion kernel takes it and exits early.
  if (J.svt == 4 && m > dh::MMAX && m < 2 * P.minimum_flank_size + J.ins_len) return 0;
  const int svS = J.sv_start, svE = J.sv_end;
  const int len1 = (int)(uint32_t)chr_len[J.chr], len2 = (int)(uint32_t)chr_len[J.chr2];
  if (J.svt == 4) {
    const int bs = std::max((int)(int32_t)(((uint64_t)(int64_t)m - (uint64_t)(int64_t)J.ins_len) / 3ull), P.minimum_flank_size);
    return clampz((long)std::min(len2, svE + bs) - std::max(0, svS - bs));
  }
  const int boundary = m;
  if (J.svt >= 5 && J.svt < 9) {
    int n = clampz((long)std::min(len1, svS + boundary) - std::max(0, svS - boundary));
    if (J.chr != J.chr2) n += clampz((long)std::min(len2, svE + boundary) - std::max(0, svE - boundary));
    return n;
  }
  const int mid = (svS + svE) / 2;
  const int sBeg = std::max(0, svS - boundary), sEnd = std::min(svS + boundary, mid);
  const int eBeg = std::max(mid + 1, svE - boundary), eEnd = std::min(len2, svE + boundary);
  switch (J.svt) {
    case 2: return (svE - svS <= P.indelsize) ? clampz((long)eEnd - sBeg) : clampz((long)sEnd - sBeg) + clampz((long)eEnd - eBeg);
    case 3: return clampz((long)eEnd - eBeg) + clampz((long)sEnd - sBeg);
    case 0: return (svE - svS > P.min_cons_window) ? clampz((long)sEnd - sBeg) + clampz((long)eEnd - eBeg)
                                                    : clampz((long)sEnd - sBeg) + clampz((long)eEnd - svS) + clampz((long)eEnd - svE);
    case 1: return (svE - svS > P.min_cons_window) ? clampz((long)sEnd - sBeg) + clampz((long)eEnd - eBeg)
                                                    : clampz((long)svS - sBeg) + clampz((long)svE - sBeg) + clampz((long)eEnd - eBeg);
    default: return 0;
  }
}

// long-read kernels: shapes beyond the short-read limits, and every junction when the orientation test
// (realign, src/split.h:564-572) is requested -- only they implement it
// Short-read junctions beyond what the sparse kernel's byte tile holds (consensus 255 .. 319 bp, windows 1281 .. 2048) have two
// homes: the packed dense kernels (~1 ms for a handful of junctions: a latency chain) and the strip kernel, whose own
// sparse passes take int16 tables.  DELLYHIP_LR_FROM_M / DELLYHIP_LR_FROM_N move the border (svt 4 keeps the insertion kernels).
static int lr_from_m() { static const int v = [] { const char* t = getenv("DELLYHIP_LR_FROM_M"); return t ? std::max(1, atoi(t)) : dh::MMAX + 1; }(); return v; }
static int lr_from_n() { static const int v = [] { const char* t = getenv("DELLYHIP_LR_FROM_N"); return t ? std::max(1, atoi(t)) : dh::NMAX + 1; }(); return v; }
bool is_lr_shape(const dellyhip_params& P, const dellyhip_junction& J, int m, int n) {
  if ((P.reserved & 1) || m > dh::MMAX || n > dh::NMAX) return true;
  return J.svt != 4 && (m >= lr_from_m() || n >= lr_from_n());
}

// Long-read workspaces are per resident wavefront and grow with the product of the batch's longest consensus and
// window (BASELINE's 10 kb x 20 kb stress shape: ~55 MB of running-max codes per wavefront): the number of resident
// wavefronts is cut so that one workspace stays below half of the free HBM (at least 1 GiB is always allowed).
uint64_t ws_budget_bytes() {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 8ull << 30;
  return std::max<uint64_t>(1ull << 30, (uint64_t)fr / 2);
}

// per-block workspace of the long-read strip kernel for consensus <= lr_m, window <= lr_n
int setup_lr_workspace(dellyhip_ctx* c, dellyhip_batch* b, int lr_m, int lr_n, int lr_cnt) {
  dh::LrArgs& R = b->lr;
  R.mcap = (lr_m + 64) & ~63;
  R.ncap = (lr_n + 64) & ~63;
  const int Q = (lr_m + 1 + dh::LRS - 1) / dh::LRS;
  R.strip_words = dh::lr_strip_words(R.ncap);
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = o; o += (bytes + 255) & ~255ull; return at; };
  take(R.mcap);                               // cons at 0
  R.off_rcons = take(R.mcap);
  R.off_ref = take(R.ncap);
  R.off_rref = take(R.ncap);
  R.off_bnd0 = take(((uint64_t)R.ncap + 128) * 4);
  R.off_bnd1 = take(((uint64_t)R.ncap + 128) * 4);
  R.off_br = take((uint64_t)dh::LR_QMAX * dh::LRS * 4);
  R.off_trF = take((uint64_t)R.mcap + R.ncap + 64);
  R.off_trR = take((uint64_t)R.mcap + R.ncap + 64);
  R.off_stack = take((uint64_t)Q * R.strip_words * 4);
  R.off_masks = take(dh::lr_masks_bytes());
  R.off_bndx = take((uint64_t)(dh::LR_TEAM_W - 1) * ((uint64_t)R.ncap + 128) * 4);
  // sparse longNeedle (sparse_needle.hpp): furthest-reaching tables for up to 256 deficit levels of this batch's
  // longest shapes (2 int16 per diagonal + 2 int32 per consensus row and level, both matrices) + the run lists
  R.sparse_bytes = 0;
  R.off_sparse = o;
  R.sparse_cost = c->sparse_cost;
  if (c->use_sparse) {
    const uint64_t ndp = ((uint64_t)lr_n + lr_m + 2 + 63) & ~63ull;
    const uint64_t per_level = 2 * ndp * 2 + 2 * (uint64_t)(lr_m + 1) * 4;
    R.sparse_bytes = 4ull * 4096 * 4 + per_level * 256;
    R.off_sparse = take(R.sparse_bytes);
  }
  R.ws_stride = o;
  // resident wavefronts: the sparse passes are latency-bound on their table loads (L2 / HBM), so as many as the
  // registers allow (156 VGPRs: 3 per SIMD) -- LDS permitting -- and the workspace budget holds
  b->lr_blocks = std::max(1, std::min(lr_cnt, c->n_cu * (c->use_sparse ? c->lr_waves : 4)));
  b->lr_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(b->lr_blocks, ws_budget_bytes() / std::max<uint64_t>(R.ws_stride, 1)));
  // teams for the dense strips (lr_dense_team_kernel): their workspaces follow lr_kernel's; the list takes two junctions per team,
  // what lr_kernel finds beyond that it sweeps itself (a batch of unalignable junctions has more wavefronts than teams)
  b->lr_teams = 0;
  R.team_state = nullptr;
  R.team_first_ws = b->lr_blocks;
  R.team_cap = 0;
  if (c->use_sparse && c->lr_teams > 0) {
    const uint64_t room = ws_budget_bytes() / std::max<uint64_t>(R.ws_stride, 1);
    const int teams = (int)std::min<uint64_t>((uint64_t)std::min(lr_cnt, c->lr_teams), room > (uint64_t)b->lr_blocks ? room - b->lr_blocks : 0);
    if (teams > 0) {
      R.team_cap = 2 * teams;
      int rc = b->lr_team_state.reserve((size_t)dh::LRT_LIST + R.team_cap + dh::LRT_DBG_INTS);   // (+ the marks of debug builds)
      if (rc) return rc;
      b->lr_teams = teams;
      R.team_state = b->lr_team_state.p;
    }
  }
  int rc = b->lr_ws.reserve((size_t)R.ws_stride * (b->lr_blocks + b->lr_teams));
  if (rc) return rc;
  R.ws = b->lr_ws.p;
  return 0;
}

// alignment-column capacity of msaEdlib for reads <= maxlen: every progressive step may add columns (insertions of the
// new read), 6 % ONT error keeps the total near 1.3 x the read length; twice the longest read + slack, at least the
// round-1 capacity, at most what the strip kernels take as a consensus
int lm_acap(int maxlen) {
  const long want = std::max<long>(4224, (2L * maxlen + 512 + 63) & ~63L);
  return (int)std::min<long>((dh::LR_MMAX + 1 + 63) & ~63, want);
}

// workspace layout of lrmsa_kernel for reads <= maxlen
void lm_layout(dh::LrMsaArgs& M, int maxlen) {
  M.acap = lm_acap(maxlen);
  M.ncap = std::min<int>((maxlen + 64) & ~63, (dh::LR_NMAX + 64) & ~63);
  M.strip_words = dh::lm_dirs_words(M.acap, M.ncap);   // capacity of the direction area (traceback-regime rectangles)
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = o; o += (bytes + 255) & ~255ull; return at; };
  take((uint64_t)dh::LM_NR * M.acap);                  // alnA at 0
  M.off_alnB = take((uint64_t)dh::LM_NR * M.acap);
  M.off_astr = take(M.acap);
  M.off_bnd = take(4ull * ((uint64_t)M.ncap + 128) * 4);
  M.off_ops = take((uint64_t)M.acap + M.ncap + 64);
  M.off_tmp = take((uint64_t)M.acap + M.ncap + 64);
  M.off_cons = take(M.acap);
  M.off_dirs = take(M.strip_words * 4);
  M.ws_stride = o;
}

// per-block workspace of the long-read insertion kernel
int setup_lri_workspace(dellyhip_ctx* c, dellyhip_batch* b, int m_max, int n_max, int cnt) {
  dh::LrInsArgs& R = b->lri;
  R.mcap = (m_max + 64) & ~63;
  R.ncap = (n_max + 64) & ~63;
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = o; o += (bytes + 255) & ~255ull; return at; };
  take(R.mcap);
  R.off_rcons = take(R.mcap);
  R.off_ref = take(R.ncap);
  R.off_rref = take(R.ncap);
  R.off_bnd = take(4ull * ((uint64_t)R.ncap + 128) * 4);
  R.off_opsL = take((uint64_t)R.mcap + R.ncap + 64);
  R.off_opsR = take((uint64_t)R.mcap + R.ncap + 64);
  R.off_tmp = take((uint64_t)R.mcap + R.ncap + 64);
  R.off_dist = take(2ull * ((uint64_t)R.ncap + 64) * 4);
  R.strip_words = dh::lm_dirs_words(R.mcap, R.ncap);   // capacity of the direction area (traceback-regime rectangles)
  R.off_dirs = take(R.strip_words * 4);
  R.ws_stride = o;
  b->lri_blocks = std::max(1, std::min(cnt, c->n_cu * 4));
  b->lri_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(b->lri_blocks, ws_budget_bytes() / std::max<uint64_t>(R.ws_stride, 1)));
  int rc = b->lri_ws.reserve((size_t)R.ws_stride * b->lri_blocks);
  if (rc) return rc;
  R.ws = b->lri_ws.p;
  return 0;
}

// workspace layout of the msaWfa kernel for reads <= maxlen; returns the per-block stride
uint64_t wfa_layout(dh::LrWfaArgs& W, int maxlen) {
  W.ncap = std::max<int>((maxlen + 64) & ~63, 64);
  // superstring / alignment columns: reads overlap almost completely (they are slices around one junction), so twice
  // the longest read plus slack holds any superstring the reference builds from them; never below the round-1 8192
  W.acap = std::min<int>(dh::WFA_ACAP_MAX, std::max<int>(8192, (2 * maxlen + 2048 + 63) & ~63));
  const int acap = W.acap;
  const int qcap = std::max(W.ncap, acap);
  W.strip_words = dh::lm_dirs_words(acap, qcap);   // capacity of the direction area
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = o; o += (bytes + 255) & ~255ull; return at; };
  take((uint64_t)(dh::LM_NR + 1) * acap);                 // alnA at 0
  W.off_alnB = take((uint64_t)(dh::LM_NR + 1) * acap);
  W.off_astr = take(acap);
  W.off_bnd = take(4ull * ((uint64_t)qcap + 128) * 4);
  W.off_ops = take(2ull * qcap + 128);
  W.off_tmp = take(2ull * qcap + 128);
  W.off_cons = take(acap);
  W.off_dirs = take(W.strip_words * 4);
  W.off_tabI = take((uint64_t)dh::WFA_KTAB * 4);
  W.off_tabJ = take((uint64_t)dh::WFA_KTAB * 4);
  W.off_diag = take(((uint64_t)acap + W.ncap + 128) * 4);
  W.off_supA = take(acap);
  W.off_supB = take(acap);
  W.off_pre = take(dh::WFA_PCAP);
  W.off_suf = take(dh::WFA_PCAP);
  W.off_edit = take((uint64_t)dh::LM_NR * dh::LM_NR * 4);
  W.ws_stride = o;
  return o;
}

// K-bins junctions by consensus length and pairs them (two junctions per wavefront, packed
// 16-bit DP).  Within a bin junctions are sorted by their approximate reference-window length
// so that partners need (almost) the same number of DP steps.  bin_count[] counts PAIRS; a
// leftover junction is paired with -1.  Junctions beyond the kernel limit go to the KMAX bin,
// where the kernel flags them with DELLYHIP_E_LIMIT.
// mode BINS_ALL: the classic layout -- every short-read junction sits in a dense bin AND (when eligible) in the sparse
// list; the dense kernels skip what split_sparse_kernel finished.  Stream slots split this in two so that the host does
// not sort junctions the dense kernels will never touch: BINS_LAZY = sparse-eligible junctions ONLY in the sparse list
// (bins hold the rest: ineligible shapes, insertions, long-read shapes); BINS_LEFTOVER = a second pass for a batch whose
// sparse kernel left junctions behind: dense bins of the sparse-eligible junctions only, nothing else.
// work_out: the list is handed back instead of being copied to b->work (the caller stages it with its other inputs).
enum { BINS_ALL = 0, BINS_LAZY = 1, BINS_LEFTOVER = 2 };
int build_bins(dellyhip_batch* b, const dellyhip_params& P, int mode = BINS_ALL, std::vector<int32_t>* work_out = nullptr) {
  b->bin_first.assign(dh::KMAX + 2, 0);
  b->bin_count.assign(dh::KMAX + 2, 0);
  std::vector<std::vector<std::pair<int, int>>> bins(dh::KMAX + 2);  // (approx n, junction)
  std::vector<std::vector<std::pair<int, int>>> qbins(7);
  b->qbin_first.assign(7, 0);
  b->qbin_count.assign(7, 0);
  b->qbin_pairs.assign(7, 0);
  std::vector<int32_t> ins, lrv, lriv, sparse, wide;
  const bool direct = b->ref_blob.p != nullptr;
  for (int i = 0; i < b->n; ++i) {
    int m = b->h_cons_len[i];
    int kk = (m + 1 + dh::WAVE - 1) / dh::WAVE;
    kk = std::max(1, std::min(kk, dh::KMAX));
    const dellyhip_junction& J = b->h_junc[i];
    if (J.svt == 4) {  // splitAlign path: own kernels, one junction per wavefront
      if (mode == BINS_LEFTOVER) continue;
      if (!b->h_win_len.empty() && is_lr_shape(P, J, m, b->h_win_len[i]) && m <= dh::LR_MMAX && b->h_win_len[i] <= dh::LR_NMAX &&
          b->lri_blocks > 0)   // (direct batches carry window lengths only beyond the short-read shapes: direct_pair)
        lriv.push_back(i);
      else ins.push_back(i);
      continue;
    }
    if (!b->h_win_len.empty() && is_lr_shape(P, J, m, b->h_win_len[i]) && m <= dh::LR_MMAX &&
        b->h_win_len[i] <= dh::LR_NMAX && b->lr_blocks > 0) {  // strip kernel (else: E_LIMIT in the short-read kernels)
      if (mode != BINS_LEFTOVER) lrv.push_back(i);
      continue;
    }
    long span = (long)J.sv_end - (long)J.sv_start;
    int approx = (J.svt == 2 && span <= P.indelsize && span >= 0) ? (int)std::min<long>(2L * m + span, 1 << 20) : 4 * m;
    if (!b->h_win_len.empty()) approx = b->h_win_len[i];
    const bool eligible = b->sr_sparse && !direct && m >= 1 && m <= dh::SPS_MMAX && approx <= dh::SPS_NMAX && approx + m + 1 <= dh::SPS_ND;
    if (eligible && mode != BINS_LEFTOVER) sparse.push_back(i);
    // (beyond the byte tile, within the dense kernels' shapes: split_sparse_wide_kernel first; like the sparse list these
    //  junctions keep their seat in a dense bin, whose kernels skip what is finished)
    if (!eligible && b->sr_sparse && b->sr_wide && !direct && m >= 1 && m <= dh::MMAX && approx <= dh::NMAX && mode != BINS_LEFTOVER) wide.push_back(i);
    if ((mode == BINS_LAZY && eligible) || (mode == BINS_LEFTOVER && !eligible)) continue;
    if (b->use_quad && !direct && m + 1 <= dh::HALF * 5 && approx <= dh::QNMAX) {
      const int kq = std::max(1, (m + 1 + dh::HALF - 1) / dh::HALF);
      qbins[kq].push_back(std::make_pair(approx, i));
    } else bins[kk].push_back(std::make_pair(approx, i));
  }
  std::vector<int32_t> work;
  work.reserve(b->n + 2 * dh::KMAX);
  // A quad bin KQ and the pair bin KP = (KQ + 1) / 2 run in ONE launch of split_quad_kernel<KQ, KP> (quad items
  // first, pair items behind them): independent bins launched one after the other would each pay the
  // ~1 ms a single DP wavefront takes, whatever their size.
  std::vector<std::vector<std::pair<int, int>>> qextra(7);
  for (int KQ = 5; KQ >= 1; --KQ) {
    const int KP = (KQ + 1) / 2;
    if (!qbins[KQ].empty() && !bins[KP].empty()) {
      qextra[KQ].swap(bins[KP]);
    }
  }
  for (int K = 1; K <= dh::KMAX; ++K) {
    auto& v = bins[K];
    std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first; });
    b->bin_first[K] = (int)work.size() / 2;
    for (size_t q = 0; q < v.size(); q += 2) {
      work.push_back(v[q].second);
      work.push_back(q + 1 < v.size() ? v[q + 1].second : -1);
    }
    b->bin_count[K] = (int)work.size() / 2 - b->bin_first[K];
  }
  for (int KQ = 1; KQ <= 5; ++KQ) {
    auto& v = qbins[KQ];
    std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first; });
    b->qbin_first[KQ] = (int)work.size();
    // All junctions of the bin are seated four per wavefront.  (Topping whole rounds of quad wavefronts up
    // with pair items -- which the kernel supports -- was measured on MI355X: 6 % faster at 40 000 junctions,
    // 9 % slower at 10 000, where neither variant fills the SIMDs; DELLYHIP_QUAD_MIX=1 enables it.)
    const int N = (int)v.size(), S = std::max(1, b->n_simd);
    int best_q = (N + 3) / 4;
    if (b->quad_mix) {
      long best_cost = -1;
      const int qmax = (N + 3) / 4;
      for (int q = (qmax / S) * S; q >= 0 && q >= qmax - 2 * S; q -= S) {   // whole rounds below qmax, and qmax itself
        for (int cand : {q, qmax}) {
          if (cand < 0 || cand > qmax) continue;
          const int rest = std::max(0, N - 4 * cand), pairs = (rest + 1) / 2;
          const long cost = 3L * ((cand + S - 1) / S) + 2L * ((pairs + S - 1) / S);   // instructions per wavefront ~ 3 : 2
          if (best_cost < 0 || cost < best_cost || (cost == best_cost && cand > best_q)) { best_cost = cost; best_q = cand; }
        }
        if (q == 0) break;
      }
    }
    size_t pos = 0;
    for (int q = 0; q < best_q; ++q)
      for (int t = 0; t < 4; ++t, ++pos) work.push_back(pos < v.size() ? v[pos].second : -1);
    b->qbin_count[KQ] = best_q;
    int np = 0;
    for (; pos < v.size(); pos += 2, ++np) {
      work.push_back(v[pos].second);
      work.push_back(pos + 1 < v.size() ? v[pos + 1].second : -1);
    }
    auto& x = qextra[KQ];   // the merged pair bin
    std::stable_sort(x.begin(), x.end(), [](const std::pair<int, int>& p1, const std::pair<int, int>& p2) { return p1.first > p2.first; });
    for (size_t q = 0; q < x.size(); q += 2, ++np) {
      work.push_back(x[q].second);
      work.push_back(q + 1 < x.size() ? x[q + 1].second : -1);
    }
    b->qbin_pairs[KQ] = np;
  }
  b->ins_first = (int)work.size();
  b->ins_count = (int)ins.size();
  work.insert(work.end(), ins.begin(), ins.end());
  if (!b->h_win_len.empty())   // largest consensus x window first: the strip kernel's wavefronts pull from this list
    std::stable_sort(lrv.begin(), lrv.end(), [&](int32_t x, int32_t y) {
      return (int64_t)b->h_cons_len[x] * b->h_win_len[x] > (int64_t)b->h_cons_len[y] * b->h_win_len[y];
    });
  b->lr_first = (int)work.size();
  b->lr_count = (int)lrv.size();
  work.insert(work.end(), lrv.begin(), lrv.end());
  b->lri_first = (int)work.size();
  b->lri_count = (int)lriv.size();
  work.insert(work.end(), lriv.begin(), lriv.end());
  b->sps_first = (int)work.size();
  b->sps_count = (int)sparse.size();
  b->sps_identity = !sparse.empty() && sparse.front() == 0 && sparse.back() == (int32_t)sparse.size() - 1;   // (ascending junction indices: then entry i is junction i)
  {
    size_t dense = 0;
    for (auto& v : bins) dense += v.size();
    for (auto& v : qbins) dense += v.size();
    for (auto& v : qextra) dense += v.size();
    b->sps_all = mode == BINS_ALL && !sparse.empty() && sparse.size() + wide.size() == dense;   // (every seat was offered to one of the two sparse kernels)
  }
  work.insert(work.end(), sparse.begin(), sparse.end());
  b->spw_first = (int)work.size();
  b->spw_count = (int)wide.size();
  work.insert(work.end(), wide.begin(), wide.end());
  if (work_out) {
    work_out->swap(work);
    return 0;
  }
  int rc = b->work.reserve(std::max<size_t>(work.size(), (size_t)3 * b->n + 2 * dh::KMAX + 64));
  if (rc) return rc;
  if (!work.empty()) HIPCHK(hipMemcpy(b->work.p, work.data(), work.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}

// consensus lengths known (h_cons_len, after an MSA stage): window lengths, routing to the short-read / strip kernels,
// strip-kernel workspaces, K bins
int route_after_msa(dellyhip_ctx* c, dellyhip_batch* b) {
  int rc;
  int lr_m = 0, lr_n = 0, lr_cnt = 0, lri_m = 0, lri_n = 0, lri_cnt = 0;
  b->h_win_len.resize(b->n);
  for (int i = 0; i < b->n; ++i) {
    const int m = b->h_cons_len[i];
    const int w = host_window_len(c->params, b->h_junc[i], m, c->chr_len);
    b->h_win_len[i] = w;
    if (is_lr_shape(c->params, b->h_junc[i], m, w) && m <= dh::LR_MMAX && w <= dh::LR_NMAX) {
      if (b->h_junc[i].svt == 4) { lri_m = std::max(lri_m, m); lri_n = std::max(lri_n, w); ++lri_cnt; }
      else { lr_m = std::max(lr_m, m); lr_n = std::max(lr_n, w); ++lr_cnt; }
    }
  }
  b->lr_blocks = 0;
  b->lri_blocks = 0;
  if (lr_cnt && (rc = setup_lr_workspace(c, b, lr_m, lr_n, lr_cnt))) return rc;
  if (lri_cnt && (rc = setup_lri_workspace(c, b, lri_m, lri_n, lri_cnt))) return rc;
  return build_bins(b, c->params);
}

}  // namespace

// long-read loop, small inversions (src/assemble.h:840-848): only the middle svSize letters of the consensus are aligned
__global__ void small_inv_apply_kernel(uint64_t* cons_off, int32_t* cons_len, const SmallInv* list, int n, uint64_t out_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SmallInv x = list[i];
  cons_off[x.j] = (uint64_t)x.j * out_stride + (uint64_t)x.offset;
  cons_len[x.j] = x.take;
}

// long-read loop, small inversions (src/assemble.h:850-853): the consensus is restored, consBp shifted
__global__ void small_inv_fix_kernel(dellyhip_result* res, const SmallInv* list, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SmallInv x = list[i];
  res[x.j].cons_len = x.full_len;
  if (res[x.j].ok) res[x.j].cons_bp += x.offset;
}

// ---- device-side compaction of the fixed-stride out blob (dellyhip_batch_fetch) ----------
// off[i] = bytes of junctions < i (consensus + "REF,ALT" + two alignment rows), off[n] = total.
// Two launches of small workgroups (round 6).  Rounds 3-5 ran ONE workgroup of 1 024 threads: it needs a compute unit with all
// sixteen wavefront slots free, and in the pipelined path it is queued while the persistent sparse kernel of the next batch holds
// every slot of the chip -- rocprofv3 showed 131 us on average (18 us alone, up to 442 us) for 31 us of work, and the slots of a
// dellyhip_stream ran their sparse kernels at 12 of 16 wavefronts per CU just to leave it room.  Workgroups of 256 threads
// (four wavefronts) slip in as wavefronts retire.
//   blob_offsets_local_kernel: workgroup g scans the lengths of records [g * BO_SPAN, + BO_SPAN) -> off[i] relative to its
//                              first record, tot[g] = its bytes
//   blob_offsets_fix_kernel:   off[i] += sum of tot[< g]; off[n] = total
constexpr int BO_SPAN = 1024;   // records per workgroup (256 threads x 4)
__host__ __device__ inline size_t blob_off_words(size_t n) { return n + 1 + (n + BO_SPAN - 1) / BO_SPAN + 1; }   // off[0 .. n], then tot[]
__global__ __launch_bounds__(256) void blob_offsets_local_kernel(const dellyhip_result* res, int n, uint64_t* off) {
  constexpr int C = BO_SPAN / 256;
  __shared__ uint64_t wsum[4];
  uint64_t* tot = off + n + 1;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int lo = (int)blockIdx.x * BO_SPAN + t * C;
  uint64_t len[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {   // (the index is clamped, not branched on: every load of the thread in flight at once)
    const int r = min(lo + i, n - 1);
    const int l0 = res[r].cons_len, l1 = res[r].allele_len, l2 = res[r].aln_len;
    const uint64_t v = (uint64_t)max(l0, 0) + (uint64_t)max(l1, 0) + 2ull * (uint64_t)max(l2, 0);
    len[i] = (lo + i < n) ? v : 0ull;
  }
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) sum += len[i];
  uint64_t inc = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint64_t at = inc - sum, total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t v = wsum[k];
    at += k < w ? v : 0ull;
    total += v;
  }
#pragma unroll
  for (int i = 0; i < C; ++i) {
    if (lo + i < n) off[lo + i] = at;
    at += len[i];
  }
  if (t == 0) tot[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void blob_offsets_fix_kernel(int n, uint64_t* off) {
  __shared__ uint64_t part[4];
  const uint64_t* tot = off + n + 1;
  const int t = threadIdx.x, g = (int)blockIdx.x, ng = (n + BO_SPAN - 1) / BO_SPAN;
  const bool last = g == ng;            // one extra workgroup writes the total
  uint64_t s = 0;
  for (int k = t; k < (last ? ng : g); k += 256) s += tot[k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) part[t >> 6] = s;
  __syncthreads();
  const uint64_t base = part[0] + part[1] + part[2] + part[3];
  if (last) {
    if (t == 0) off[n] = base;
    return;
  }
  if (base)
    for (int i = g * BO_SPAN + t; i < min(n, (g + 1) * BO_SPAN); i += 256) off[i] += base;
}
// off must hold blob_off_words(n) words
static inline void launch_blob_offsets(hipStream_t s, const dellyhip_result* res, int n, uint64_t* off) {
  if (n <= 0) {
    (void)hipMemsetAsync(off, 0, sizeof(uint64_t), s);
    return;
  }
  const int ng = (n + BO_SPAN - 1) / BO_SPAN;
  hipLaunchKernelGGL(blob_offsets_local_kernel, dim3(ng), dim3(256), 0, s, res, n, off);
  hipLaunchKernelGGL(blob_offsets_fix_kernel, dim3(ng + 1), dim3(256), 0, s, n, off);
}
// one wavefront per junction: its three pieces, back to back, at out + off[i]
__global__ void blob_gather_kernel(const dellyhip_result* res, const uint8_t* blob, const uint64_t* off, uint8_t* out, int n, uint64_t cap) {
  const int lane = threadIdx.x;
  if (off[n] > cap) return;   // (dellyhip_batch_fetch_begin: the size is not known on the host when this is queued)
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const dellyhip_result R = res[i];
    uint8_t* dst = out + off[i];
    const uint64_t src[3] = {R.cons_off, R.allele_off, R.aln_off};
    const int len[3] = {max(R.cons_len, 0), max(R.allele_len, 0), 2 * max(R.aln_len, 0)};
    for (int k = 0; k < 3; ++k) {
      for (int q = lane; q < len[k]; q += dh::WAVE) dst[q] = blob[src[k] + q];
      dst += len[k];
    }
  }
}

// dellyhip_batch_fetch_begin: what dellyhip_batch_fetch does on the host after its downloads, done on the device with the CALLER's
// pinned host memory as the destination.  Record i leaves as the first 36 lanes' dwords of one wavefront store, its blob offsets
// rebased to the compacted layout (rebase_offsets below) and `reserved` cleared; then the compacted bytes in 16-byte pieces.
// status[0] = blob bytes, status[1] = flags (1: they do not fit `cap`, nothing of the blob written; 2: *lrt_error is set).
__global__ __launch_bounds__(256) void fetch_out_kernel(const dellyhip_result* res, const uint64_t* off, int n, const uint8_t* compact,
                                                        dellyhip_result* h_res, uint8_t* h_blob, uint64_t cap, uint64_t* status,
                                                        const int32_t* lrt_error) {
  constexpr int RW = (int)(sizeof(dellyhip_result) / 4);
  static_assert(sizeof(dellyhip_result) % 4 == 0 && RW <= dh::WAVE, "one record = one wavefront store");
  constexpr int W_CONS_LEN = (int)(offsetof(dellyhip_result, cons_len) / 4), W_ALLELE_LEN = (int)(offsetof(dellyhip_result, allele_len) / 4),
                W_ALN_LEN = (int)(offsetof(dellyhip_result, aln_len) / 4), W_CONS_OFF = (int)(offsetof(dellyhip_result, cons_off) / 4),
                W_ALLELE_OFF = (int)(offsetof(dellyhip_result, allele_off) / 4), W_ALN_OFF = (int)(offsetof(dellyhip_result, aln_off) / 4),
                W_RESERVED = (int)(offsetof(dellyhip_result, reserved) / 4);
  const uint64_t used = off[n];
  const bool fits = used <= cap;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
  const int lane = (int)(threadIdx.x & 63);
  for (uint64_t i = tid >> 6; i < (uint64_t)n; i += nthr >> 6) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(res + i);
    uint32_t w = lane < RW ? src[lane] : 0u;
    const uint64_t l0 = (uint64_t)max(__builtin_amdgcn_readlane((int)w, W_CONS_LEN), 0), l1 = (uint64_t)max(__builtin_amdgcn_readlane((int)w, W_ALLELE_LEN), 0),
                   l2 = 2ull * (uint64_t)max(__builtin_amdgcn_readlane((int)w, W_ALN_LEN), 0);
    const uint64_t at = off[i];
    const uint64_t o0 = l0 ? at : 0, o1 = l1 ? at + l0 : 0, o2 = l2 ? at + l0 + l1 : 0;
    w = lane == W_CONS_OFF ? (uint32_t)o0 : lane == W_CONS_OFF + 1 ? (uint32_t)(o0 >> 32) : w;
    w = lane == W_ALLELE_OFF ? (uint32_t)o1 : lane == W_ALLELE_OFF + 1 ? (uint32_t)(o1 >> 32) : w;
    w = lane == W_ALN_OFF ? (uint32_t)o2 : lane == W_ALN_OFF + 1 ? (uint32_t)(o2 >> 32) : w;
#ifndef DH_LR_TIMING
    w = lane == W_RESERVED ? 0u : w;
#endif
    if (lane < RW) reinterpret_cast<uint32_t*>(h_res + i)[lane] = w;
  }
  if (fits) {
    // 16-byte pieces between a byte-wise head and tail; the host side gives `compact` the destination's alignment mod 16 (ADVICE r05:
    // an unaligned out_blob fell back to byte-wise stores over PCIe for the whole blob, silently)
    const uintptr_t ah = reinterpret_cast<uintptr_t>(h_blob) & 15, ac = reinterpret_cast<uintptr_t>(compact) & 15;
    const uint64_t head = (ah == ac) ? min(used, (uint64_t)((16 - ah) & 15)) : used;
    const uint64_t n16 = (used - head) >> 4;
    for (uint64_t q = tid; q < head; q += nthr) h_blob[q] = compact[q];
    for (uint64_t q = tid; q < n16; q += nthr) reinterpret_cast<uint4*>(h_blob + head)[q] = reinterpret_cast<const uint4*>(compact + head)[q];
    for (uint64_t q = head + (n16 << 4) + tid; q < used; q += nthr) h_blob[q] = compact[q];
  }
  if (tid == 0) {
    status[0] = used;
    status[1] = (fits ? 0u : 1u) | ((lrt_error && *lrt_error) ? 2u : 0u);
  }
}

namespace {

// Pinned host memory, grown geometrically and kept.  Released blocks go to a process-wide free list like the device
// blocks of DevPool (hipHostMalloc / hipHostFree synchronise the device).
struct PinPool {
  struct Block { void* p; size_t bytes; int device; };
  std::mutex mu;
  std::vector<Block> free_list;
  static PinPool& get() { static PinPool* P = new PinPool(); return *P; }   // (never destroyed: the runtime may be gone at exit)
  void* take(size_t want, size_t* got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> g(mu);
      size_t best = free_list.size();
      for (size_t i = 0; i < free_list.size(); ++i)
        if (free_list[i].device == dev && free_list[i].bytes >= want && free_list[i].bytes <= 4 * want + (1u << 20) &&
            (best == free_list.size() || free_list[i].bytes < free_list[best].bytes))
          best = i;
      if (best != free_list.size()) {
        Block b = free_list[best];
        free_list.erase(free_list.begin() + (long)best);
        *got = b.bytes;
        return b.p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    *got = want;
    return p;
  }
  void give(void* p, size_t bytes) {
    const int dev = DevPool::device_of(p, true);   // (the device the block was pinned for, whatever the current one is)
    std::lock_guard<std::mutex> g(mu);
    free_list.push_back(Block{p, bytes, dev});
  }
  size_t trim(int dev) {
    std::vector<Block> mine;
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size();)
        if (free_list[i].device == dev) { mine.push_back(free_list[i]); free_list.erase(free_list.begin() + (long)i); }
        else ++i;
    }
    size_t bytes = 0;
    for (auto& b : mine) { (void)hipHostFree(b.p); bytes += b.bytes; }
    return bytes;
  }
};
template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t bytes_ = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { release(); }
  int reserve(size_t count) {
    if (p && n >= count) return 0;
    release();
    const size_t want = count + count / 4 + 64;
    p = static_cast<T*>(PinPool::get().take(want * sizeof(T), &bytes_));
    if (!p) return fail(DELLYHIP_E_NOMEM, "hipHostMalloc", hipErrorOutOfMemory);
    n = bytes_ / sizeof(T);
    return 0;
  }
  void release() {
    if (p) PinPool::get().give(p, bytes_);
    p = nullptr;
    n = 0;
    bytes_ = 0;
  }
};

// the batch's own pinned block for the downloaded consensus lengths (n ints, grown and kept)
static int batch_pin_len(dellyhip_batch* b) {
  const size_t want = (size_t)std::max(b->n, 1);
  if (b->own_pin_len && b->own_pin_count >= want) return 0;
  if (b->own_pin_len) PinPool::get().give(b->own_pin_len, b->own_pin_bytes);
  b->own_pin_len = static_cast<int32_t*>(PinPool::get().take((want + want / 4 + 64) * sizeof(int32_t), &b->own_pin_bytes));
  if (!b->own_pin_len) { b->own_pin_bytes = b->own_pin_count = 0; return fail(DELLYHIP_E_NOMEM, "hipHostMalloc", hipErrorOutOfMemory); }
  b->own_pin_count = b->own_pin_bytes / sizeof(int32_t);
  return 0;
}

// Staging arena of a stream slot: one pinned host block and one device block with the SAME layout.  Everything a batch
// uploads (junction records, offsets, sequence bytes, work lists) is appended to the host block, the batch's device pointers
// are borrowed from the device block at the same offsets, and ONE asynchronous copy moves the used prefix.
struct Arena {
  PinBuf<uint8_t> h;
  DevBuf<uint8_t> d;
  size_t used = 0;
  int begin(size_t cap) {
    used = 0;
    int rc = h.reserve(cap);
    if (!rc) rc = d.reserve_grow(cap);
    return rc;
  }
  // -> host pointer of the appended range (the caller may still fill it); *dev = its device twin
  template <typename T>
  T* put(const T* src, size_t count, size_t pad_bytes, T** dev) {
    used = (used + 15) & ~(size_t)15;
    const size_t bytes = count * sizeof(T);
    if (used + bytes + pad_bytes > h.n || used + bytes + pad_bytes > d.n) return nullptr;
    T* hp = reinterpret_cast<T*>(h.p + used);
    if (src && bytes) memcpy(hp, src, bytes);
    *dev = reinterpret_cast<T*>(d.p + used);
    used += bytes + pad_bytes;
    return hp;
  }
};

struct UploadOpts {
  dellyhip_batch* recycle = nullptr;   // reuse this batch object and its device allocations
  Arena* arena = nullptr;              // stage the inputs (else: one allocation + synchronous copy per array)
  bool lazy = false;                   // stream slot: host routing of what split_sparse_kernel leaves behind is deferred
  hipStream_t up = nullptr;            // the staged inputs travel on this stream (else the context's); up_done is recorded
  hipEvent_t up_done = nullptr;        // behind the copy and the context's stream waits for it
  bool zero_copy_blob = false;         // a PINNED seq_blob is read by the copy engine in place instead of being staged (dellyhip_stream_zero_copy)
};

// per-batch host state back to "freshly constructed", device allocations kept
void batch_reset(dellyhip_batch* b) {
  b->ever_run = false; b->probe_mode = 0; b->n = 0; b->n_seq = 0; b->with_msa = 0; b->want_alignment = 0;
  b->h_junc.clear(); b->h_cons_len.clear(); b->h_win_len.clear();
  b->out_stride = 0;
  b->out_cons_cap = dh::OUT_CONS_CAP; b->out_allele_cap = dh::OUT_ALLELE_CAP; b->out_aln_cap = dh::OUT_ALN_CAP;
  b->bin_first.clear(); b->bin_count.clear(); b->qbin_first.clear(); b->qbin_count.clear(); b->qbin_pairs.clear();
  b->ins_first = b->ins_count = 0; b->sps_all = false; b->sps_identity = false; b->early_count = 0; b->early_done = false; b->sps_first = b->sps_count = 0; b->spw_first = b->spw_count = 0;
  b->lr_first = b->lr_count = b->lr_blocks = 0; b->lri_first = b->lri_count = b->lri_blocks = 0;
  b->wfa_items = 0; b->wfa_pair_grid = 1; b->wfa_count = b->wfa_blocks = 0; b->small_inv_n = 0;
  b->lm_hbuf_half = 0; b->lm_pair_grid = 1; b->lm_items = b->lm_blocks = 0;
  b->msa_big_grid = 0; b->lazy = 0; b->lazy_pending = false;
  b->ms_split = b->ms_msa = b->ms_dp = b->ms_dp_last = 0; b->launches = 0; b->pending = false;
  b->ref_blob.release(); b->ref_off.release(); b->ref_len.release();
}

}  // namespace

// stream slots: the compaction also emits the records as the caller sees them -- blob offsets rebased to the compact blob,
// transient kernel state (result.reserved) cleared -- so that the host only copies
__global__ void blob_gather_records_kernel(const dellyhip_result* res, const uint8_t* blob, const uint64_t* off, uint8_t* out,
                                           dellyhip_result* rec_out, int n, uint64_t* hdr_used, int32_t* hdr_left, const int32_t* left_counter) {
  // the record travels as 36 dwords, one per lane (one coalesced load and store); the three offsets, `reserved` and the
  // padding word are replaced on the way
  constexpr int RW = (int)(sizeof(dellyhip_result) / 4);
  constexpr int W_CONS = (int)(offsetof(dellyhip_result, cons_off) / 4), W_ALLELE = (int)(offsetof(dellyhip_result, allele_off) / 4),
                W_ALN = (int)(offsetof(dellyhip_result, aln_off) / 4), W_RESERVED = (int)(offsetof(dellyhip_result, reserved) / 4),
                W_PAD = (int)(offsetof(dellyhip_result, ref_len) / 4) + 1;
  static_assert(sizeof(dellyhip_result) % 4 == 0 && RW <= dh::WAVE && W_PAD + 1 == W_CONS, "record layout");
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0 && hdr_used) {   // head of the block the records travel in (StreamHeader): one D2H copy for both
    *hdr_used = off[n];
    hdr_left[0] = left_counter ? *left_counter : 0;
    hdr_left[1] = 0;
  }
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const dellyhip_result& R = res[i];
    uint32_t v = lane < RW ? reinterpret_cast<const uint32_t*>(res + i)[lane] : 0u;
    uint64_t at = off[i];
    uint8_t* dst = out + at;
    const uint64_t src[3] = {R.cons_off, R.allele_off, R.aln_off};
    const int len[3] = {max(R.cons_len, 0), max(R.allele_len, 0), 2 * max(R.aln_len, 0)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      for (int q = lane; q < len[k]; q += dh::WAVE) dst[q] = blob[src[k] + q];
      dst += len[k];
    }
    const uint64_t o0 = len[0] ? at : 0, o1 = len[1] ? at + len[0] : 0, o2 = len[2] ? at + len[0] + len[1] : 0;
    v = lane == W_CONS ? (uint32_t)o0 : lane == W_CONS + 1 ? (uint32_t)(o0 >> 32) : v;
    v = lane == W_ALLELE ? (uint32_t)o1 : lane == W_ALLELE + 1 ? (uint32_t)(o1 >> 32) : v;
    v = lane == W_ALN ? (uint32_t)o2 : lane == W_ALN + 1 ? (uint32_t)(o2 >> 32) : v;
    v = (lane == W_RESERVED || lane == W_PAD) ? 0u : v;
    if (lane < RW) reinterpret_cast<uint32_t*>(rec_out + i)[lane] = v;
  }
}

extern "C" {

const char* dellyhip_last_error(void) { return g_err.c_str(); }

void dellyhip_abi_info(int32_t out[4]) {
  out[0] = DELLYHIP_VERSION;
  out[1] = (int32_t)sizeof(dellyhip_params);
  out[2] = (int32_t)sizeof(dellyhip_junction);
  out[3] = (int32_t)sizeof(dellyhip_result);
}

void dellyhip_default_params_sr(dellyhip_params* p) {
  *p = dellyhip_params{5, -4, -10, -1, 2, 13, 1000, 100, 0.95f, 0};
}
void dellyhip_default_params_lr(dellyhip_params* p) {
  *p = dellyhip_params{5, -4, -10, -1, 3, 100, 10000, 1000, 0.9f, 0};
}

static int create_ctx(const dellyhip_params* params, int device, dellyhip_ctx** out, hipStream_t borrowed);
int dellyhip_create(const dellyhip_params* params, int device, dellyhip_ctx** out) { return create_ctx(params, device, out, nullptr); }

// `borrowed`: the context launches on that stream and does not own it (slots of a dellyhip_stream)
static int create_ctx(const dellyhip_params* params, int device, dellyhip_ctx** out, hipStream_t borrowed) {
  if (!params || !out) return fail(DELLYHIP_E_ARG, "null argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(DELLYHIP_E_NODEVICE, "no HIP device", e);
  if (device < 0 || device >= ndev) return fail(DELLYHIP_E_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DELLYHIP_E_NODEVICE, "device is not gfx950 (kernels are built for MI355X only)");
  dellyhip_ctx* c = new dellyhip_ctx();
  c->device = device;
  c->params = *params;
  c->n_cu = prop.multiProcessorCount;
  c->chrs = std::make_shared<ChrTable>();
  c->chrs->device = device;
  if (const char* t = getenv("DELLYHIP_QUAD")) c->use_quad = atoi(t) != 0;  // tuning / test knobs
  if (const char* t = getenv("DELLYHIP_SPARSE")) c->use_sparse = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_SR_SPARSE")) c->sr_sparse = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_LR_WAVES")) c->lr_waves = std::max(1, std::min(8, atoi(t)));
  if (const char* t = getenv("DELLYHIP_WFA_LDS_SEED")) c->wfa_lds_seed = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_MYERS_BAND")) c->myers_band = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_LRC_WAVES")) c->lrc_waves = std::max(1, std::min(4, atoi(t)));
  if (const char* t = getenv("DELLYHIP_LR_TEAMS")) c->lr_teams = std::max(0, std::min(256, atoi(t)));
  if (const char* t = getenv("DELLYHIP_LR_TEAMS_SERIAL")) c->lr_team_serial = atoi(t) ? 1 : 0;
  if (const char* t = getenv("DELLYHIP_SPS_WAVES")) c->sps_waves = std::max(1, std::min(20, atoi(t)));
  if (const char* t = getenv("DELLYHIP_SPARSE_COST")) c->sparse_cost = std::max(1, atoi(t));
  if (const char* t = getenv("DELLYHIP_QUAD_MIX")) c->quad_mix = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_MSA_ONLY")) c->msa_only = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_MSA_WAVES")) c->msa_waves = std::max(1, std::min(16, atoi(t)));
  if (const char* t = getenv("DELLYHIP_MSA_TEAM")) c->msa_team = atoi(t);
  if (const char* t = getenv("DELLYHIP_MSA_PAIR")) c->msa_pair = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_SR_WIDE")) c->sr_wide = atoi(t) != 0;
  if (const char* t = getenv("DELLYHIP_MSA_TMAX")) c->msa_tmax = std::max(0, std::min(atoi(t), (int)dh::TMAXC));  // tuning / test knob
  if (borrowed) {
    c->stream = borrowed;
    c->owns_stream = false;
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return fail(DELLYHIP_E_RUNTIME, "hipStreamCreate", e);
    }
  }
  *out = c;
  return 0;
}

static int create_shared_on(dellyhip_ctx* share_with, const dellyhip_params* params, dellyhip_ctx** out, hipStream_t borrowed) {
  if (!share_with || !out) return fail(DELLYHIP_E_ARG, "null argument");
  dellyhip_ctx* c = nullptr;
  int rc = create_ctx(params ? params : &share_with->params, share_with->device, &c, borrowed);
  if (rc) return rc;
  c->chrs = share_with->chrs;
  *out = c;
  return 0;
}
int dellyhip_create_shared(dellyhip_ctx* share_with, const dellyhip_params* params, dellyhip_ctx** out) {
  return create_shared_on(share_with, params, out, nullptr);
}

int dellyhip_host_register(dellyhip_ctx* c, void* p, uint64_t bytes) {
  if (!c || !p || !bytes) return fail(DELLYHIP_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e != hipSuccess) return fail(DELLYHIP_E_RUNTIME, "hipHostRegister", e);
  return 0;
}

int dellyhip_host_unregister(dellyhip_ctx* c, void* p) {
  if (!c || !p) return fail(DELLYHIP_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  (void)hipDeviceSynchronize();   // (a copy into the range may still be in flight)
  hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) return fail(DELLYHIP_E_RUNTIME, "hipHostUnregister", e);
  return 0;
}

uint64_t dellyhip_trim_memory(dellyhip_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->spw_scratch.release();   // (lazily re-allocated by the next launch of split_sparse_wide_kernel)
  c->spw_blocks = 0;
  return (uint64_t)DevPool::get().trim(c->device) + (uint64_t)PinPool::get().trim(c->device);
}

void dellyhip_destroy(dellyhip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& hs : c->host_streams) {
    if (hs) dellyhip_stream_destroy(hs);
    hs = nullptr;
  }
  c->chrs.reset();   // (the last context of a table frees the chromosomes)
  c->d_chr_ptr.release();
  c->d_chr_len.release();
  c->scratch.release();
  c->spw_scratch.release();
  c->counters.release();
  if (c->serial_ev) (void)hipEventDestroy(c->serial_ev);
  for (auto q : c->lr_aux_all)
    if (q != c->lr_aux) (void)hipStreamDestroy(q);   // (the one that was picked serves every context of the device until the process ends)
  if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int dellyhip_set_chromosome(dellyhip_ctx* c, int32_t chr, const char* seq, int64_t len) {
  if (!c || chr < 0 || len < 0 || (!seq && len)) return fail(DELLYHIP_E_ARG, "bad chromosome");
  HIPCHK(hipSetDevice(c->device));
  uint8_t* d = nullptr;   // allocate and fill the new buffer first: a failure leaves the old chromosome in place
  hipError_t e = dh::dev_alloc((void**)&d, (size_t)std::max<int64_t>(len, 1));
  if (e != hipSuccess) return fail(DELLYHIP_E_NOMEM, "hipMalloc(chromosome)", e);
  if (len) {
    e = hipMemcpy(d, seq, (size_t)len, hipMemcpyHostToDevice);
    if (e != hipSuccess) { dh::dev_free(d); return fail(DELLYHIP_E_RUNTIME, "H2D chromosome", e); }
  }
  {
    std::lock_guard<std::mutex> g(c->chrs->mu);
    ChrTable& T = *c->chrs;
    if ((size_t)chr >= T.dev.size()) {
      T.dev.resize(chr + 1, nullptr);
      T.len.resize(chr + 1, 0);
    }
    if (T.dev[chr]) {
      (void)hipDeviceSynchronize();   // (a kernel of any context sharing the table may still read the old copy)
      dh::dev_free(T.dev[chr]);
    }
    T.dev[chr] = d;
    T.len[chr] = len;
    ++T.version;
  }
  refresh_chr(c);
  return 0;
}

void dellyhip_batch_free(dellyhip_ctx* c, dellyhip_batch* b) {
  if (!b) return;
  if (c) (void)hipSetDevice(c->device);
  if (b->pending && c) (void)hipStreamSynchronize(c->stream);
  if (b->fetch_pending && b->fetch_ev && c) (void)hipEventSynchronize(b->fetch_ev);   // (its kernels write into the caller's memory and read the batch's)
  // (a batch run on a caller's stream, or whose run failed after the teams were launched: the batch's workspace and team state
  //  must not be released while lr_dense_team_kernel may still poll them)
  if (b->lr_aux_used && c && c->lr_aux) (void)hipStreamSynchronize(c->lr_aux);
  b->junc.release(); b->seq_blob.release(); b->seq_off.release(); b->cons_off.release();
  b->cons_len.release(); b->res.release(); b->out_blob.release(); b->work.release();
  b->ref_blob.release(); b->ref_off.release(); b->ref_len.release(); b->msa_ws.release(); b->msa_big_ws.release(); b->lm_hbuf.release(); b->lr_ws.release(); b->blob_off.release(); b->blob_compact.release(); b->lm_edit.release(); b->lm_pair_first.release(); b->lm_ws.release(); b->lri_ws.release(); b->wfa_list.release(); b->wfa_ws.release(); b->early_list.release(); b->msa_order.release(); b->wfa_pair_first.release(); b->wfa_edit.release(); b->wfa_seeds.release(); b->wfa_pair_ws.release(); b->wfa_next.release(); b->small_inv.release(); b->lr_team_state.release();
  if (b->own_pin_len) PinPool::get().give(b->own_pin_len, b->own_pin_bytes);
  b->own_pin_len = nullptr;
  if (b->fetch_status) PinPool::get().give(b->fetch_status, b->fetch_status_bytes);
  b->fetch_status = nullptr;
  if (b->fetch_ev) (void)hipEventDestroy(b->fetch_ev);
  if (b->lr_fork) (void)hipEventDestroy(b->lr_fork);
  if (b->lr_join) (void)hipEventDestroy(b->lr_join);
  if (b->lri_fork) (void)hipEventDestroy(b->lri_fork);
  if (b->lri_join) (void)hipEventDestroy(b->lri_join);
  for (auto e : b->ev) if (e) (void)hipEventDestroy(e);
  for (auto e : b->ev_free) (void)hipEventDestroy(e);
  if (b->len_ev) (void)hipEventDestroy(b->len_ev);
  delete b;
}

static int batch_upload_impl(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                             const uint64_t* seq_off, uint64_t n_seq, int with_msa, int want_alignment,
                             dellyhip_batch** out, const UploadOpts* opts = nullptr) {
  if (!c || !out || n < 0 || (n && (!junc || !seq_off))) return fail(DELLYHIP_E_ARG, "bad batch arguments");
  HIPCHK(hipSetDevice(c->device));
  refresh_chr(c);
  if (n_seq && (!seq_off || (seq_off[n_seq] && !seq_blob))) return fail(DELLYHIP_E_ARG, "null sequence blob / offsets");
  for (uint64_t i = 0; i < n_seq; ++i)
    if (seq_off[i + 1] < seq_off[i] || seq_off[i + 1] - seq_off[i] > 0x7fffffffull)
      return fail(DELLYHIP_E_ARG, "seq_off is not monotonic");
  for (int i = 0; i < n; ++i) {
    const dellyhip_junction& J = junc[i];
    if (J.n_seq < 0 || J.seq_first > n_seq || J.seq_first + (uint64_t)J.n_seq > n_seq) return fail(DELLYHIP_E_ARG, "junction sequence range");
    if (!with_msa && J.n_seq != 1) return fail(DELLYHIP_E_ARG, "align_consensus needs exactly one sequence per junction");
    if (J.chr < 0 || J.chr2 < 0 || (size_t)J.chr >= c->chr_dev.size() || (size_t)J.chr2 >= c->chr_dev.size() ||
        !c->chr_dev[J.chr] || !c->chr_dev[J.chr2])
      return fail(DELLYHIP_E_ARG, "junction refers to a chromosome that was not uploaded");
    // The window arithmetic of _initBreakpoint (src/tags.h:151-172) clamps against 0 and target_len only on one side
    // each; a breakpoint outside its chromosome makes the reference's substr() calls throw / read out of bounds,
    // so such records are malformed input here too.  Types 0-4 are intra-chromosomal (src/tags.h:22-25).
    if (J.sv_start < 0 || J.sv_end < 0 || (int64_t)J.sv_start > c->chr_len[J.chr] || (int64_t)J.sv_end > c->chr_len[J.chr2])
      return fail(DELLYHIP_E_ARG, "junction coordinates outside the chromosome");
    if (J.svt >= 0 && J.svt <= 4 && J.chr != J.chr2) return fail(DELLYHIP_E_ARG, "svt 0-4 with chr != chr2");
  }
  const bool recycle = opts && opts->recycle;
  Arena* arena = opts ? opts->arena : nullptr;
  dellyhip_batch* b = recycle ? opts->recycle : new dellyhip_batch();
  if (recycle) batch_reset(b);
  b->n = n;
  b->n_seq = n_seq;
  b->with_msa = with_msa;
  b->want_alignment = want_alignment;
  b->use_quad = c->use_quad;
  b->quad_mix = c->quad_mix;
  b->sr_sparse = c->sr_sparse;
  b->sr_wide = c->sr_wide != 0;
  b->n_simd = c->n_cu * 4;
  b->h_junc.assign(junc, junc + n);
  int lr_m = 0, lr_n = 0, lr_cnt = 0, lri_m = 0, lri_n = 0, lri_cnt = 0;
  if (!with_msa) {  // |svRefStr| per junction; long-read shapes get larger output slots and a workspace
    b->h_win_len.resize(n);
    for (int i = 0; i < n; ++i) {
      const int m = (int)(seq_off[junc[i].seq_first + 1] - seq_off[junc[i].seq_first]);
      const int w = host_window_len(c->params, junc[i], m, c->chr_len);
      b->h_win_len[i] = w;
      if (is_lr_shape(c->params, junc[i], m, w) && m <= dh::LR_MMAX && w <= dh::LR_NMAX) {
        if (junc[i].svt == 4) { lri_m = std::max(lri_m, m); lri_n = std::max(lri_n, w); ++lri_cnt; }
        else { lr_m = std::max(lr_m, m); lr_n = std::max(lr_n, w); ++lr_cnt; }
      }
    }
  }
  int msa_maxlen = 1;   // longest read of the batch (with_msa == 2)
  if (with_msa == 2) {  // long-read MSA: consensus / window lengths are only bounded at upload time
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < junc[i].n_seq; ++k)
        msa_maxlen = std::max<int>(msa_maxlen, (int)std::min<uint64_t>(seq_off[junc[i].seq_first + k + 1] - seq_off[junc[i].seq_first + k], 1u << 20));
    // consensus <= alignment columns <= lm_acap(longest read); exact alleles exist only for svEnd - svStart <= indelsize
    // (src/split.h:606), where the window is contiguous: |svRefStr| <= 2 |consensus| + indelsize (src/split.h:116)
    const int acap = lm_acap(msa_maxlen);
    b->out_cons_cap = (acap + 15) & ~15;
    const long win = std::min<long>(dh::LR_NMAX, 2L * acap + std::max(c->params.indelsize, 0));
    b->out_allele_cap = (int)((acap + win + 8 + 15) & ~15L);
    b->out_aln_cap = 2 * (int)((acap + (long)dh::LR_NMAX + 8 + 15) & ~15L);
  }
  if (with_msa == 1) {
    // msa(): a consensus is at most the alignment's columns; reads of one junction overlap the breakpoint, so twice the
    // longest read bounds it in practice (longer: DELLYHIP_E_LIMIT for that junction).  Batches of <= 128 bp reads keep
    // the short-read slot sizes.
    b->msa_plan = dh::msa_prepare(b->h_junc, seq_off);
    const int cc = std::min<int>(dh::msa_big::LCAP, (2 * b->msa_plan.maxlen + 64 + 15) & ~15);
    if (cc > dh::OUT_CONS_CAP) {
      const long win2 = std::min<long>(dh::LR_NMAX, 2L * cc + std::max(c->params.indelsize, 0));   // contiguous window (src/split.h:116)
      const long winmax = std::min<long>(dh::LR_NMAX, std::max<long>(win2, 4L * cc));                // two windows (:117)
      b->out_cons_cap = cc;
      b->out_allele_cap = (int)((cc + win2 + 8 + 15) & ~15L);
      b->out_aln_cap = 2 * (int)((cc + winmax + 8 + 15) & ~15L);
    }
  }
  if (lr_cnt || lri_cnt) {
    const int mm = std::max(lr_m, lri_m), nn = std::max(lr_n, lri_n);
    b->out_cons_cap = std::max<int>(dh::OUT_CONS_CAP, (mm + 16) & ~15);
    b->out_allele_cap = std::max<int>(dh::OUT_ALLELE_CAP, (mm + nn + 8 + 15) & ~15);
    b->out_aln_cap = std::max<int>(dh::OUT_ALN_CAP, 2 * ((mm + nn + 8 + 15) & ~15));
  }
  b->out_stride = (uint64_t)b->out_cons_cap + b->out_allele_cap + (want_alignment ? b->out_aln_cap : 0);
  b->out_stride = (b->out_stride + 15) & ~15ull;
  uint64_t blob_bytes = n_seq ? seq_off[n_seq] : 0;
  int rc = 0;
  auto bail = [&](int r) {
    if (!recycle) dellyhip_batch_free(c, b);
    return r;
  };
  hipError_t e;
  // one array of the batch onto the device: staged (arena) or allocation + synchronous copy
  auto push = [&](auto& dst, const auto* src, size_t count, size_t pad_elems, const char* what) -> int {
    typedef typename std::remove_reference<decltype(*dst.p)>::type T;
    if (arena) {
      T* dev = nullptr;
      if (!arena->put<T>(src, count, pad_elems * sizeof(T), &dev)) return fail(DELLYHIP_E_RUNTIME, "staging arena overflow");
      dst.borrow(dev, count + pad_elems);
      return 0;
    }
    int r = dst.alloc(std::max<size_t>(count + pad_elems, 1));
    if (r) return r;
    if (count) {
      hipError_t ee = hipMemcpy(dst.p, src, count * sizeof(T), hipMemcpyHostToDevice);
      if (ee != hipSuccess) return fail(DELLYHIP_E_RUNTIME, what, ee);
    }
    return 0;
  };
  if (arena) {
    // everything push() appends below: records, offsets, bytes, per-junction arrays, work lists (<= 3 n + bins), pair lists
    const size_t cap = (size_t)n * (sizeof(dellyhip_junction) + 8 + 4 + 16 + 12 + 8) + (n_seq + 1) * 8 + blob_bytes + 64 + 4096;
    if ((rc = arena->begin(cap))) return bail(rc);
  }
  if ((rc = push(b->junc, junc, (size_t)n, 0, "H2D junctions"))) return bail(rc);
  // The sequence bytes are three quarters of what a batch uploads and the largest part of the host's staging time (one memcpy into
  // the pinned arena).  A caller that keeps them in pinned memory (dellyhip_host_register / hipHostMalloc) and has said so
  // (dellyhip_stream_zero_copy) is spared the copy: the bytes get their place at the END of the arena's device block and travel
  // from the caller's buffer by a copy of their own (below, behind the arena's).
  bool blob_in_place = false;
  if (arena && opts->zero_copy_blob && blob_bytes >= 4096) {
    void *d0 = nullptr, *d1 = nullptr;
    blob_in_place = hipHostGetDevicePointer(&d0, const_cast<char*>(seq_blob), 0) == hipSuccess &&
                    hipHostGetDevicePointer(&d1, const_cast<char*>(seq_blob) + blob_bytes - 1, 0) == hipSuccess;
    if (!blob_in_place) (void)hipGetLastError();   // (pageable memory: staged as usual)
  }
  if (!blob_in_place && (rc = push(b->seq_blob, reinterpret_cast<const uint8_t*>(seq_blob), (size_t)blob_bytes, 64, "H2D sequences"))) return bail(rc);   // (+64: the bit-vector kernels fetch pattern bytes 32 at a time)
  if ((rc = push(b->seq_off, seq_off, (size_t)n_seq + 1, 0, "H2D offsets"))) return bail(rc);
  const bool cons_len_staged = arena && !with_msa;   // (given consensus: the lengths travel with the other inputs)
  if (recycle) {
    if ((!cons_len_staged && (rc = b->cons_len.reserve_grow(std::max(n, 1)))) || (rc = b->res.reserve_grow(std::max(n, 1))) ||
        (rc = b->out_blob.reserve_grow(std::max<uint64_t>((uint64_t)n * b->out_stride, 1))))
      return bail(rc);
  } else {
    if (!cons_len_staged && (rc = b->cons_len.alloc(std::max(n, 1)))) return bail(rc);
    if ((rc = b->res.alloc(std::max(n, 1)))) return bail(rc);
    if ((rc = b->out_blob.alloc(std::max<uint64_t>((uint64_t)n * b->out_stride, 1)))) return bail(rc);
  }
  // (memsets go onto the context's stream: hipMemset on the null stream returns before the fill is done and is not ordered
  //  with a hipStreamNonBlocking stream -- a kernel launched there could be overwritten by the fill.  Staged uploads stay
  //  asynchronous: every later operation of the batch is enqueued on the same stream.)
  e = hipMemsetAsync(b->res.p, 0, std::max(n, 1) * sizeof(dellyhip_result), c->stream);
  if (e == hipSuccess && !arena) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return bail(fail(DELLYHIP_E_RUNTIME, "memset results", e));
  b->h_cons_len.resize(n);
  if (!with_msa) {
    std::vector<uint64_t> coff(n);
    for (int i = 0; i < n; ++i) {
      coff[i] = seq_off[junc[i].seq_first];
      b->h_cons_len[i] = (int32_t)(seq_off[junc[i].seq_first + 1] - seq_off[junc[i].seq_first]);
    }
    if ((rc = push(b->cons_off, coff.data(), (size_t)n, 0, "H2D cons_off"))) return bail(rc);
    if (arena) {
      if ((rc = push(b->cons_len, b->h_cons_len.data(), (size_t)n, 4, "H2D cons_len"))) return bail(rc);
    } else if (n) {
      e = hipMemcpy(b->cons_len.p, b->h_cons_len.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
      if (e != hipSuccess) return bail(fail(DELLYHIP_E_RUNTIME, "H2D cons_len", e));
    }
    if (lr_cnt && (rc = setup_lr_workspace(c, b, lr_m, lr_n, lr_cnt))) return bail(rc);
    if (lri_cnt && (rc = setup_lri_workspace(c, b, lri_m, lri_n, lri_cnt))) return bail(rc);
    if (arena) {
      std::vector<int32_t> work;
      b->lazy = (opts->lazy && b->sr_sparse) ? 1 : 0;
      if ((rc = build_bins(b, c->params, b->lazy ? BINS_LAZY : BINS_ALL, &work))) return bail(rc);
      if ((rc = push(b->work, work.data(), work.size(), 64, "H2D work list"))) return bail(rc);
    } else if ((rc = build_bins(b, c->params))) return bail(rc);
  } else {
    // consensus is produced on the device at out_blob + i*stride
    std::vector<uint64_t> coff(n);
    for (int i = 0; i < n; ++i) coff[i] = (uint64_t)i * b->out_stride;
    if ((rc = push(b->cons_off, coff.data(), (size_t)n, 0, "H2D cons_off"))) return bail(rc);
    if (with_msa == 2) {
      // msaEdlib: all-pairs work list + per-block workspace sized from the longest read
      std::vector<int32_t> pf(n + 1, 0);
      const int maxlen = msa_maxlen;
      bool long_pairs = false;   // a junction with two reads beyond the rows of one bit-vector pass: strip passes need a byte workspace
      for (int i = 0; i < n; ++i) {
        const int N = std::max(0, std::min(junc[i].n_seq, (int)dh::LM_NR));
        pf[i + 1] = pf[i] + ((junc[i].n_seq <= dh::LM_NR && junc[i].svt != 4) ? N * (N - 1) / 2 : 0);   // (insertions: msaWfa's own scores)
        int nlong = 0;
        for (int k = 0; k < junc[i].n_seq; ++k)
          nlong += (seq_off[junc[i].seq_first + k + 1] - seq_off[junc[i].seq_first + k] > (uint64_t)dh::MYERS_ROWS) ? 1 : 0;
        long_pairs |= nlong >= 2;
      }
      b->lm_hbuf_half = long_pairs ? (((uint64_t)maxlen + 16 + 255) & ~255ull) : 0;
      b->lm_maxlen = maxlen;
      b->lm_items = pf[n];
      if ((rc = push(b->lm_pair_first, pf.data(), (size_t)n + 1, 0, "H2D pair list")) ||
          (rc = b->lm_edit.reserve(std::max<size_t>((size_t)n * dh::LM_NR * dh::LM_NR, 1))))
        return bail(rc);
      dh::LrMsaArgs& M = b->lm;
      lm_layout(M, maxlen);
      b->lm_blocks = std::max(1, std::min(n, c->n_cu * 4 * c->lrc_waves));
      b->lm_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(b->lm_blocks, ws_budget_bytes() / std::max<uint64_t>(M.ws_stride, 1)));
      if ((rc = b->lm_ws.reserve((size_t)M.ws_stride * b->lm_blocks))) return bail(rc);
      M.ws = b->lm_ws.p;
      b->lm_pair_grid = std::max(1, std::min(b->lm_items, c->n_cu * 16));
      if (b->lm_hbuf_half && (rc = b->lm_hbuf.reserve((size_t)2 * b->lm_hbuf_half * b->lm_pair_grid))) return bail(rc);
      // insertions: msaWfa kernel
      std::vector<int32_t> wl;
      for (int i = 0; i < n; ++i)
        if (junc[i].svt == 4) wl.push_back(i);
      b->wfa_count = (int)wl.size();
      if (b->wfa_count) {
        wfa_layout(b->wfa, maxlen);
        b->wfa_blocks = std::max(1, std::min(b->wfa_count, c->n_cu * 4 * c->lrc_waves));
        b->wfa_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(b->wfa_blocks, ws_budget_bytes() / std::max<uint64_t>(b->wfa.ws_stride, 1)));
        if ((rc = push(b->wfa_list, wl.data(), wl.size(), 0, "H2D wfa list")) || (rc = b->wfa_ws.reserve((size_t)b->wfa.ws_stride * b->wfa_blocks))) return bail(rc);
        e = hipMemsetAsync(b->wfa_ws.p, 0, (size_t)b->wfa.ws_stride * b->wfa_blocks, c->stream);   // k-mer tables start (and are kept) all zero
        if (e == hipSuccess && !arena) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return bail(fail(DELLYHIP_E_RUNTIME, "memset wfa workspace", e));
        b->wfa.ws = b->wfa_ws.p;
        // pairwise scores of every insertion junction: work list + workspace of wfa_pairs_kernel
        std::vector<int32_t> wpf(n + 1, 0);
        for (int i = 0; i < n; ++i) {
          const int N = junc[i].n_seq;
          wpf[i + 1] = wpf[i] + ((junc[i].svt == 4 && N >= 2 && N <= dh::LM_NR) ? N * (N - 1) / 2 : 0);
        }
        b->wfa_items = wpf[n];
        dh::WfaPairArgs& WP = b->wfa_pairs;
        WP.ncap = b->wfa.ncap;
        WP.acap = b->wfa.acap;
        WP.off_tabJ = (uint64_t)dh::WFA_KTAB * 4;
        WP.off_diag = 2 * WP.off_tabJ;
        WP.off_hb = WP.off_diag + (((uint64_t)2 * maxlen + 64 + 63) & ~63ull) * 4;
        WP.hb_half = (maxlen > dh::MYERS_ROWS) ? (((uint64_t)maxlen + 16 + 255) & ~255ull) : 0;
        WP.ws_stride = (WP.off_hb + 2 * WP.hb_half + 255) & ~255ull;
        b->wfa_pair_grid = std::max(1, std::min(b->wfa_items, c->n_cu * 16));
        b->wfa_pair_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(b->wfa_pair_grid, ws_budget_bytes() / WP.ws_stride));
        if (b->wfa_items > 0) {
          if ((rc = push(b->wfa_pair_first, wpf.data(), (size_t)n + 1, 0, "H2D wfa pair list")) || (rc = b->wfa_edit.reserve((size_t)n * dh::LM_NR * dh::LM_NR)) ||
              (rc = b->wfa_pair_ws.reserve((size_t)WP.ws_stride * b->wfa_pair_grid)) || (rc = b->wfa_next.reserve(1)))
            return bail(rc);
          e = hipMemsetAsync(b->wfa_pair_ws.p, 0, (size_t)WP.ws_stride * b->wfa_pair_grid, c->stream);   // k-mer tables start (and are kept) all zero
          if (e == hipSuccess) e = hipMemsetAsync(b->wfa_edit.p, 0, (size_t)n * dh::LM_NR * dh::LM_NR * sizeof(int32_t), c->stream);
          if (e == hipSuccess && !arena) e = hipStreamSynchronize(c->stream);
          if (e != hipSuccess) return bail(fail(DELLYHIP_E_RUNTIME, "memset wfa pair workspace", e));
          WP.pair_first = b->wfa_pair_first.p;
          WP.edit = b->wfa_edit.p;
          // the diagonal seeding with its tables in LDS (wfa_seed_kernel): four ints per pair for the distance kernel
          b->wfa_max_rows = 0;
          for (int i = 0; i < n; ++i)
            if (junc[i].svt == 4 && junc[i].n_seq >= 2 && junc[i].n_seq <= dh::LM_NR) b->wfa_max_rows = std::max(b->wfa_max_rows, junc[i].n_seq - 1);
          if (c->wfa_lds_seed && (rc = b->wfa_seeds.reserve((size_t)4 * b->wfa_items))) return bail(rc);
          WP.ws = b->wfa_pair_ws.p;
          WP.next = b->wfa_next.p;
          WP.n_junc = n;
          WP.n_items = b->wfa_items;
        }
      }
    } else {
      const dh::MsaPlan& mp = b->msa_plan;
      b->msa_team = dh::msa_team_waves(n, c->n_cu * c->msa_waves, c->msa_team);
      b->msa_grid = dh::msa_team_grid(n, c->n_cu * c->msa_waves, b->msa_team);
      b->msa_stride = dh::msa_team_stride(mp.nmax, b->msa_team);
      if ((rc = b->msa_ws.reserve(std::max<uint64_t>(1, b->msa_stride * (uint64_t)b->msa_grid)))) return bail(rc);
      // msa_big: as many resident wavefronts as junctions are expected there; a few stand by for the unpredictable
      // case (a node of the standard instance growing beyond its 512 columns)
      b->msa_big_grid = mp.big_count > 0 ? std::min(mp.big_count, c->n_cu * 2) : std::min(std::max(n, 1), 8);
      if ((rc = b->msa_big_ws.reserve(std::max<uint64_t>(1, mp.big_ws_stride * (uint64_t)b->msa_big_grid)))) return bail(rc);
      {
        // Longest-processing-time first: a junction of 20 reads costs ~15x one of 5 (pairs x length^2 + merges x length^2), and
        // a long one that starts last IS the tail of the launch.  Junctions are independent (src/shortpe.h:183-197: every task
        // writes only its own record), so the order the kernel takes them in is free.  Skipped when all cost the same.
        std::vector<std::pair<uint64_t, int32_t>> cost(n);
        uint64_t lo = ~0ull, hi = 0;
        for (int i = 0; i < n; ++i) {
          const uint64_t nr = (uint64_t)std::max(junc[i].n_seq, 0);
          const uint64_t len = nr ? (seq_off[junc[i].seq_first + nr] - seq_off[junc[i].seq_first]) / nr : 0;
          const uint64_t cst = (nr * (nr - (nr ? 1 : 0)) / 2 + 6 * nr) * len * len;
          cost[i] = {cst, i};
          lo = std::min(lo, cst);
          hi = std::max(hi, cst);
        }
        b->msa_ordered = n > 1 && hi > lo + lo / 8 && !getenv("DELLYHIP_MSA_NO_ORDER");
        if (b->msa_ordered) {
          std::stable_sort(cost.begin(), cost.end(), [](const std::pair<uint64_t, int32_t>& x, const std::pair<uint64_t, int32_t>& y) { return x.first > y.first; });
          std::vector<int32_t> order(n);
          for (int i = 0; i < n; ++i) order[i] = cost[i].second;
          if ((rc = push(b->msa_order, order.data(), order.size(), 0, "H2D msa order"))) return bail(rc);
        }
      }
      if (c->sr_sparse && !(c->params.reserved & 1)) {   // (long-read parameters route every junction to the strip kernel)
        std::vector<int32_t> el;
        for (int i = 0; i < n; ++i)
          if (junc[i].svt != 4) el.push_back(i);
        b->early_count = (int)el.size();
        if (b->early_count && (rc = push(b->early_list, el.data(), el.size(), 0, "H2D early list"))) return bail(rc);
        // every junction goes through split_sparse_kernel first: a stream slot routes what it leaves behind (usually
        // nothing) when the results are collected, not in the middle of the run
        b->lazy = (opts && opts->lazy && b->early_count == n && n > 0) ? 1 : 0;
      }
    }
  }
  if (arena && arena->used) {
    hipStream_t up = (opts->up && opts->up_done) ? opts->up : c->stream;
    const size_t staged = arena->used;
    if (blob_in_place) {   // (reserved last: the arena's own copy ends in front of it)
      uint8_t* dev = nullptr;
      if (!arena->put<uint8_t>(nullptr, (size_t)blob_bytes, 64, &dev)) return bail(fail(DELLYHIP_E_RUNTIME, "staging arena overflow"));
      b->seq_blob.borrow(dev, (size_t)blob_bytes + 64);
    }
    e = hipMemcpyAsync(arena->d.p, arena->h.p, staged, hipMemcpyHostToDevice, up);
    if (e == hipSuccess && blob_in_place) e = hipMemcpyAsync(b->seq_blob.p, seq_blob, (size_t)blob_bytes, hipMemcpyHostToDevice, up);
    if (e == hipSuccess && up != c->stream) {
      e = hipEventRecord(opts->up_done, up);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, opts->up_done, 0);
    }
    if (e != hipSuccess) return bail(fail(DELLYHIP_E_RUNTIME, "H2D staged inputs", e));
  }
  *out = b;
  return 0;
}

int dellyhip_batch_upload(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                          const uint64_t* seq_off, uint64_t n_seq, int with_msa, dellyhip_batch** out) {
  if ((with_msa & ~16) < 0 || (with_msa & ~16) > 2) return fail(DELLYHIP_E_ARG, "with_msa");
  int rc = batch_upload_impl(c, n, junc, seq_blob, seq_off, n_seq, with_msa & ~16, 0, out);
  if (!rc && (with_msa & 16)) (*out)->probe_mode = 1;
  return rc;
}

static inline double now_s();
int dellyhip_batch_run(dellyhip_ctx* c, dellyhip_batch* b, void* stream) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "null argument");
  // DELLYHIP_TRACE_RUN=1: a run that keeps the calling thread longer than 0.2 ms says where (stderr); =2: every run.  (Round 6 used it to
  // show that the late first enqueue behind a sync -- CHANGELOG round 6 -- is not spent in here.)
  static const bool trace_run = getenv("DELLYHIP_TRACE_RUN") != nullptr;
  const double tr0 = trace_run ? now_s() : 0;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (b->n == 0) return 0;
  double tr1 = 0, tr2 = 0, tr3 = 0;
  struct TraceRun { bool on; double t0; double *a, *b_, *c_; ~TraceRun() { if (on) { double e = now_s(); if (e - t0 > 2e-4 || getenv("DELLYHIP_TRACE_RUN")[0] == '2') fprintf(stderr, "batch_run: %.3f ms on the host (events %.3f, serialisation %.3f, up to the split stage %.3f)\n", (e - t0) * 1e3, (*a - t0) * 1e3, (*b_ - *a) * 1e3, (*c_ - *b_) * 1e3); } } } trace_guard{trace_run, tr0, &tr1, &tr2, &tr3};
  // a fetch begun and not ended reads the results this run overwrites: same stream -> ordered already
  if (b->fetch_pending && b->fetch_ev && b->fetch_stream != s) HIPCHK(hipStreamWaitEvent(s, b->fetch_ev, 0));
  b->run_stream = s;
  int rc;
  hipEvent_t e3[4];
  bool ev1_done = false;
  // Every event recorded on the stream is a barrier packet the next kernel sits behind for 5 - 7 us (kernel trace of the headline
  // arrangement: 21 us between the last kernel of a step and the first of the next, four records).  A run records what its
  // timing needs and nothing else: start of the MSA stage only when there is one (slot 0 stays empty), start of the split stage,
  // end of the sparse kernel, end of the run; the context's serialisation event is recorded when a run on ANOTHER stream has to
  // wait for this one, not after every run.
  for (int q = 0; q < 4; ++q) {
    if (q == 0 && !b->with_msa) { e3[q] = nullptr; b->ev.push_back(nullptr); continue; }
    if (!b->ev_free.empty()) { e3[q] = b->ev_free.back(); b->ev_free.pop_back(); }
    else HIPCHK(hipEventCreate(&e3[q]));
    b->ev.push_back(e3[q]);
  }
  b->mid = e3[3];
  if (trace_run) tr1 = now_s();
  if (c->serial_valid && c->serial_stream != s) {   // (same stream: ordered already)
    if (!c->serial_ev) HIPCHK(hipEventCreateWithFlags(&c->serial_ev, hipEventDisableTiming));
    if (hipEventRecord(c->serial_ev, c->serial_stream) == hipSuccess) {
      HIPCHK(hipStreamWaitEvent(s, c->serial_ev, 0));
    } else {   // (the caller's stream of the previous run is gone)
      (void)hipGetLastError();
      HIPCHK(hipDeviceSynchronize());
    }
  }
  if (trace_run) tr2 = now_s();
  if (e3[0]) HIPCHK(hipEventRecord(e3[0], s));
  // with_msa == 0: junction_setup treats res[j].status / sr_support as input from an MSA stage; without one they
  // must be zero on EVERY run, or a junction flagged in run 1 takes the "prior status" branch in run 2
  if (!b->with_msa && b->ever_run) b->zero_res_pending = true;   // (the upload zeroed them for the first run; run_split zeroes them together with its counters: launch_zero2)
  if (b->with_msa == 2) {
    // msaEdlib (src/assemble.h:383-473): all-pairs bit-vector distances, then one wavefront per junction
    if (b->lm_items > 0) {
      dh::PairArgs pa{b->junc.p, b->seq_blob.p, b->seq_off.p, b->lm_pair_first.p, b->n, b->lm_items, dh::LM_NR, b->lm_edit.p,
                      b->lm_hbuf.p, b->lm_hbuf_half, 0, 0, 0};
      if (c->myers_band) {   // banded passes, several pairs per wavefront (myers_band.hpp): sized from the batch's longest read
        pa.band_g = dh::mb_group(b->lm_maxlen);
        pa.band_k = dh::mb_band(b->lm_maxlen);
        pa.band_wl = dh::mb_lanes(pa.band_k);
      }
      int pgrid = b->lm_pair_grid;
      if (pa.band_g >= 2) {   // groups of band_g pairs per wavefront and step: no more workgroups than stay resident (15 per CU by LDS), whole rounds
        const int groups = (b->lm_items + pa.band_g - 1) / pa.band_g;
        pgrid = std::max(1, std::min(groups, c->n_cu * 12));
        const int rounds = (groups + pgrid - 1) / pgrid;
        pgrid = (groups + rounds - 1) / rounds;
      }
      hipLaunchKernelGGL(dh::myers_pairs_kernel, dim3(pgrid), dim3(dh::WAVE), 0, s, pa);
      HIPCHK(hipGetLastError());
    }
    dh::LrMsaArgs M = b->lm;
    M.junc = b->junc.p;
    M.seq_blob = b->seq_blob.p;
    M.seq_off = b->seq_off.p;
    M.p = c->params;
    M.res = b->res.p;
    M.out_blob = b->out_blob.p;
    M.out_stride = b->out_stride;
    M.out_cons_cap = b->out_cons_cap;
    M.cons_len = b->cons_len.p;
    M.edit = b->lm_edit.p;
    M.n_work = b->n;
    hipLaunchKernelGGL(dh::lrmsa_kernel, dim3(b->lm_blocks), dim3(dh::WAVE), 0, s, M);
    HIPCHK(hipGetLastError());
    if (b->wfa_count > 0) {   // insertions: msaWfa with the reference anchors of src/assemble.h:855-856
      if ((rc = ensure_chr_table(c))) return rc;
      dh::LrWfaArgs W = b->wfa;
      W.junc = b->junc.p; W.seq_blob = b->seq_blob.p; W.seq_off = b->seq_off.p;
      W.chr_seq = c->d_chr_ptr.p; W.chr_len = c->d_chr_len.p;
      W.p = c->params; W.res = b->res.p; W.out_blob = b->out_blob.p; W.out_stride = b->out_stride;
      W.out_cons_cap = b->out_cons_cap; W.cons_len = b->cons_len.p;
      W.work_list = b->wfa_list.p; W.n_work = b->wfa_count; W.use_anchors = 1;
      W.edit_all = nullptr;
      if (b->wfa_items > 0) {   // the pairwise scores of all junctions first: throughput work for the whole chip
        dh::WfaPairArgs WP = b->wfa_pairs;
        WP.junc = b->junc.p; WP.seq_blob = b->seq_blob.p; WP.seq_off = b->seq_off.p;
        HIPCHK(hipMemsetAsync(WP.next, 0, sizeof(uint32_t), s));
        WP.seeds = nullptr;
        if (c->wfa_lds_seed && b->wfa_seeds.p && b->wfa_max_rows > 0) {
          HIPCHK(hipMemsetAsync(WP.next, 0, sizeof(uint32_t), s));
          dh::WfaSeedArgs SA{b->junc.p, b->seq_blob.p, b->seq_off.p, WP.pair_first, b->wfa_list.p, b->wfa_count, b->wfa_max_rows, WP.ncap, WP.acap, b->wfa_seeds.p, WP.next};
          hipLaunchKernelGGL(dh::wfa_seed_kernel, dim3(std::max(1, std::min(b->wfa_count * b->wfa_max_rows, c->n_cu * 2))), dim3(dh::WAVE), 0, s, SA);
          HIPCHK(hipGetLastError());
          HIPCHK(hipMemsetAsync(WP.next, 0, sizeof(uint32_t), s));
          WP.seeds = b->wfa_seeds.p;
        }
        WP.band_g = WP.band_k = WP.band_wl = 0;
        if (c->myers_band) {   // banded distances, several pairs per wavefront and pass (myers_band.hpp): sized from the batch's longest read
          WP.band_g = dh::mb_group(b->lm_maxlen);
          WP.band_k = dh::mb_band(b->lm_maxlen);
          WP.band_wl = dh::mb_lanes(WP.band_k);
        }
        hipLaunchKernelGGL(dh::wfa_pairs_kernel, dim3(b->wfa_pair_grid), dim3(dh::WAVE), 0, s, WP);
        HIPCHK(hipGetLastError());
        W.edit_all = b->wfa_edit.p;
      }
      hipLaunchKernelGGL(dh::lrwfa_kernel, dim3(b->wfa_blocks), dim3(dh::WAVE), 0, s, W);
      HIPCHK(hipGetLastError());
    }
    if ((rc = batch_pin_len(b))) return rc;
    HIPCHK(hipMemcpyAsync(b->own_pin_len, b->cons_len.p, b->n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(b->h_cons_len.data(), b->own_pin_len, (size_t)b->n * sizeof(int32_t));
    // "take care of small inversions" (src/assemble.h:840-848): align only the middle svSize letters
    {
      std::vector<SmallInv> si;
      for (int i = 0; i < b->n; ++i) {
        const dellyhip_junction& J = b->h_junc[i];
        const int32_t svSize = J.sv_end - J.sv_start, m = b->h_cons_len[i];
        if ((J.svt == 0 || J.svt == 1) && svSize < m && m > 0) {
          const int32_t off = (int32_t)(((size_t)m - (size_t)svSize) / 2);
          if (off < 0 || off > m) continue;
          const int32_t take = std::min<int32_t>(std::max(svSize, 0), m - off);
          si.push_back(SmallInv{i, m, off, take});
          b->h_cons_len[i] = take;
        }
      }
      b->small_inv_n = (int)si.size();
      if (b->small_inv_n) {   // one upload + one launch for all of them
        if ((rc = b->small_inv.reserve(si.size()))) return rc;
        HIPCHK(hipMemcpy(b->small_inv.p, si.data(), si.size() * sizeof(SmallInv), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(small_inv_apply_kernel, dim3((b->small_inv_n + 63) / 64), dim3(64), 0, s, b->cons_off.p, b->cons_len.p, b->small_inv.p,
                           b->small_inv_n, b->out_stride);
        HIPCHK(hipGetLastError());
      }
    }
    if ((rc = route_after_msa(c, b))) return rc;
  } else if (b->with_msa) {
    if ((rc = ensure_scratch(c))) return rc;
    HIPCHK(hipMemsetAsync(c->counters.p, 0, 32 * sizeof(int32_t), s));
    dh::MsaArgs ma{};
    ma.junc = b->junc.p;
    ma.seq_blob = b->seq_blob.p;
    ma.seq_off = b->seq_off.p;
    ma.p = c->params;
    ma.res = b->res.p;
    ma.out_blob = b->out_blob.p;
    ma.out_stride = b->out_stride;
    ma.cons_len = b->cons_len.p;
    ma.ws = b->msa_ws.p;
    ma.ws_stride = b->msa_stride;
    ma.out_cons_cap = b->out_cons_cap;
    ma.big_counter = c->counters.p + 9;
    ma.n_work = b->n;
    ma.work_counter = c->counters.p;
    ma.defer_counter = c->counters.p + 8;
    ma.tmax = dh::msa_tmax(c->params, c->msa_tmax);
    ma.pair = c->msa_pair;
    ma.order = b->msa_ordered ? b->msa_order.p : nullptr;
    if ((rc = dh::msa_launch(ma, b->msa_grid, b->msa_plan.nmax, s, b->msa_big_ws.p, b->msa_plan.big_ws_stride, b->msa_big_grid,
                             b->msa_plan.big_nmax, b->msa_team)))
      return fail(rc, "msa_launch");
    HIPCHK(hipGetLastError());
    if (c->msa_only) {   // profiling: the records keep what the MSA kernels wrote (tools/msa_phases.py)
      HIPCHK(hipEventRecord(e3[1], s));
      HIPCHK(hipEventRecord(e3[3], s));
      HIPCHK(hipEventRecord(e3[2], s));
      c->serial_valid = true;
      c->serial_stream = s;
      b->last = e3[2];
      b->pending = true;
      b->launches++;
      b->ever_run = true;
      return 0;
    }
    // consensus lengths decide the K bin of the split kernel
    int32_t* len_dst = (b->lazy && b->pin_cons_len) ? b->pin_cons_len : nullptr;
    if (!len_dst) {
      if ((rc = batch_pin_len(b))) return rc;
      len_dst = b->own_pin_len;
    }
    HIPCHK(hipMemcpyAsync(len_dst, b->cons_len.p, b->n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (b->lazy) {
      // stream slot: every junction goes through split_sparse_kernel; what it leaves behind (counters[31]) is routed to the
      // dense kernels when the results are collected (finish_lazy) -- no host synchronisation inside the run
      HIPCHK(hipEventRecord(e3[1], s));
      ev1_done = true;
      if ((rc = launch_early_sparse(c, b, s))) return rc;
    } else if (b->early_count > 0) {
      // the sparse kernel needs no routing (it reads the lengths on the device and leaves what it cannot take to the
      // dense kernels): it runs while the host bins the batch from the downloaded lengths
      if (!b->len_ev) HIPCHK(hipEventCreateWithFlags(&b->len_ev, hipEventDisableTiming));
      HIPCHK(hipEventRecord(b->len_ev, s));
      HIPCHK(hipEventRecord(e3[1], s));
      ev1_done = true;
      if ((rc = launch_early_sparse(c, b, s))) return rc;
      HIPCHK(hipEventSynchronize(b->len_ev));
    } else {
      HIPCHK(hipStreamSynchronize(s));
    }
    if (!b->lazy) memcpy(b->h_cons_len.data(), b->own_pin_len, (size_t)b->n * sizeof(int32_t));
    if (!b->lazy && (rc = route_after_msa(c, b))) return rc;   // (a consensus beyond 319 bp goes to the strip kernel)
  }
  if (!ev1_done) HIPCHK(hipEventRecord(e3[1], s));
  if (trace_run) tr3 = now_s();
  if (!(b->lazy && b->with_msa == 1) && (rc = run_split(c, b, s, b->ref_blob.p != nullptr))) return rc;
  b->lazy_pending = b->lazy != 0;
  if (b->with_msa == 2 && b->small_inv_n > 0) {
    hipLaunchKernelGGL(small_inv_fix_kernel, dim3((b->small_inv_n + 63) / 64), dim3(64), 0, s, b->res.p, b->small_inv.p, b->small_inv_n);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(e3[2], s));
  c->serial_valid = true;
  c->serial_stream = s;
  b->last = e3[2];
  b->pending = true;
  b->lr_failed = false;
  b->launches++;
  b->ever_run = true;
  return 0;
}

int dellyhip_batch_sync(dellyhip_ctx* c, dellyhip_batch* b) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "null argument");
  if (!b->pending)   // (sticky until the next run: a second sync, a fetch, a gather or a stream collect after the error must not hand the records out)
    return b->lr_failed ? fail(DELLYHIP_E_RUNTIME, "a long-read team of wavefronts gave up waiting: the junctions it was to sweep are not refined") : 0;
  if (getenv("DELLYHIP_TRACE_SYNC")) {   // (debugging aid: a batch that does not finish says how far its stream got)
    const auto t0 = std::chrono::steady_clock::now();
    while (hipEventQuery(b->last) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    if (hipEventQuery(b->last) == hipErrorNotReady)
      fprintf(stderr, "dellyhip_batch_sync: not finished after 5 s; long-read fork event (after the team state memsets, before lr_kernel): %s\n",
              !b->lr_fork ? "none" : hipEventQuery(b->lr_fork) == hipSuccess ? "reached" : "NOT reached");
    (void)hipGetLastError();
  }
  HIPCHK(hipEventSynchronize(b->last));
  for (size_t q = 0; q + 3 < b->ev.size(); q += 4) {
    float a = 0, d = 0, e = 0;
    if (b->ev[q]) HIPCHK(hipEventElapsedTime(&a, b->ev[q], b->ev[q + 1]));   // (no MSA stage: no start event)
    HIPCHK(hipEventElapsedTime(&d, b->ev[q + 1], b->ev[q + 2]));
    HIPCHK(hipEventElapsedTime(&e, b->ev[q + 1], b->ev[q + 3]));
    b->ms_msa += a;
    b->ms_split += d;
    b->ms_dp += e;
  }
  for (auto e : b->ev) if (e) b->ev_free.push_back(e);
  b->ev.clear();
  b->pending = false;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DELLYHIP_E_RUNTIME, "kernel execution", e);
  if (b->lr_teams > 0 && b->lr_team_state.p) {
    // a team of lr_dense_team_kernel that gave up waiting (for its list entry, for lr_kernel, inside its own pipe) leaves junctions
    // with the "not refined, status 0" record lr_kernel wrote when it deferred them: the batch must not be handed out as good
    int32_t flag = 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(&flag, b->lr_team_state.p + dh::LRT_ERROR, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
      b->lr_failed = true;
      return fail(DELLYHIP_E_RUNTIME, "a long-read team of wavefronts gave up waiting: the junctions it was to sweep are not refined");
    }
  }
  return 0;
}

// src/split.h:606-624 on the host, for records produced with DELLYHIP_COMPACT_ALLELES (see dellyhip.h)
int64_t dellyhip_recut_alleles(const dellyhip_params* P, const dellyhip_junction* J, const dellyhip_result* R, const char* cons,
                               const char* seq, int64_t chr_len, char* out, uint64_t cap) {
  if (!P || !J || !R || !out) return fail(DELLYHIP_E_ARG, "null argument");
  if (R->allele_len >= 0) return 0;
  if (!cons || !seq) return fail(DELLYHIP_E_ARG, "null argument");
  const int64_t total = -(int64_t)R->allele_len;
  const int64_t nr = (int64_t)R->r_end - R->r_start, na = (int64_t)R->c_end - R->c_start;
  if (!R->ok || (J->svt != 2 && J->svt != 4) || nr < 0 || na < 0 || nr + na + 1 != total || R->c_start < 1 || R->r_start < 1 || R->c_end - 1 > R->cons_len)
    return fail(DELLYHIP_E_ARG, "dellyhip_recut_alleles: the record does not describe compact alleles");
  if ((uint64_t)total > cap) return fail(DELLYHIP_E_ARG, "dellyhip_recut_alleles: out too small");
  // window start = bp.svStartBeg (src/tags.h:151-172; alignConsensus src/split.h:650-655)
  const int32_t boundary = (J->svt == 4) ? std::max((int32_t)(((size_t)R->cons_len - (size_t)(int64_t)J->ins_len) / 3), P->minimum_flank_size) : R->cons_len;
  const int64_t beg = std::max(0, J->sv_start - boundary);
  const int64_t r0 = beg + R->r_start - 1;
  if (r0 < 0 || r0 + nr > chr_len) return fail(DELLYHIP_E_ARG, "dellyhip_recut_alleles: the reference allele leaves the chromosome");
  for (int64_t i = 0; i < nr; ++i) {
    const unsigned char ch = (unsigned char)seq[r0 + i];
    out[i] = (char)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);   // boost::to_upper_copy of _getSVRef (src/split.h:116-119)
  }
  out[nr] = ',';
  memcpy(out + nr + 1, cons + (R->c_start - 1), (size_t)na);
  return total;
}

// the same over a batch: out_off[i] .. out_off[i + 1] = "REF,ALT" of record i (empty where it has no compact alleles)
int64_t dellyhip_recut_alleles_batch(const dellyhip_params* P, int32_t n, const dellyhip_junction* J, const dellyhip_result* R, const char* blob,
                                     const char* const* chr_seq, const int64_t* chr_len, int32_t n_chr, char* out, uint64_t cap, uint64_t* out_off) {
  if (!P || n < 0 || (n && (!J || !R || !blob || !chr_seq || !chr_len || !out_off))) return fail(DELLYHIP_E_ARG, "null argument");
  uint64_t at = 0;
  for (int32_t i = 0; i < n; ++i) {
    out_off[i] = at;
    if (R[i].allele_len >= 0) continue;
    if (J[i].chr < 0 || J[i].chr >= n_chr) return fail(DELLYHIP_E_ARG, "dellyhip_recut_alleles_batch: chromosome index outside the table");
    const int64_t got = dellyhip_recut_alleles(P, &J[i], &R[i], blob + R[i].cons_off, chr_seq[J[i].chr], chr_len[J[i].chr], out + at, cap - at);
    if (got < 0) return got;
    at += (uint64_t)got;
  }
  if (n) out_off[n] = at;
  return (int64_t)at;
}

int dellyhip_batch_device_results(dellyhip_ctx* c, dellyhip_batch* b, void** dptr, uint64_t* bytes) {
  if (!c || !b || !dptr || !bytes) return fail(DELLYHIP_E_ARG, "null argument");
  *dptr = b->res.p;
  *bytes = (uint64_t)b->n * sizeof(dellyhip_result);
  return 0;
}

int dellyhip_batch_sparse_left(dellyhip_ctx* c, dellyhip_batch* b, int32_t* left) {
  if (!c || !b || !left) return fail(DELLYHIP_E_ARG, "null argument");
  *left = 0;
  int rc = dellyhip_batch_sync(c, b);
  if (rc || !c->counters.p || !b->ever_run) return rc;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy(left, c->counters.p + 31, sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}

int dellyhip_batch_msa_stats(dellyhip_ctx* c, dellyhip_batch* b, int32_t out[4]) {
  if (!c || !b || !out) return fail(DELLYHIP_E_ARG, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  int rc = dellyhip_batch_sync(c, b);
  if (rc || !c->counters.p || !b->ever_run) return rc;
  HIPCHK(hipSetDevice(c->device));
  int32_t v[2] = {0, 0};
  HIPCHK(hipMemcpy(v, c->counters.p + 8, sizeof(v), hipMemcpyDeviceToHost));
  out[0] = v[0];
  out[1] = v[1];
  out[2] = b->msa_team;
  out[3] = b->msa_grid;
  return 0;
}

int dellyhip_batch_lr_team_stats(dellyhip_ctx* c, dellyhip_batch* b, int32_t out[4]) {
  if (!c || !b || !out) return fail(DELLYHIP_E_ARG, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  int rc = dellyhip_batch_sync(c, b);
  if ((rc && !b->lr_failed) || !b->ever_run || b->lr_teams <= 0 || !b->lr_team_state.p) return rc;
  HIPCHK(hipSetDevice(c->device));   // (a team that gave up: out[] is filled -- out[3] = 1 -- and the sync error returned)
  int32_t v[4];
  HIPCHK(hipMemcpy(v, b->lr_team_state.p, sizeof(v), hipMemcpyDeviceToHost));
#ifdef DH_LR_TEAM_DEBUG
  {
    std::vector<int32_t> all((size_t)dh::LRT_LIST + b->lr.team_cap + 8 * (size_t)b->lr_teams);
    HIPCHK(hipMemcpy(all.data(), b->lr_team_state.p, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    fprintf(stderr, "lr teams: count %d taken %d done %d error %d | list", all[0], all[1], all[2], all[3]);
    for (int i = 0; i < std::min(b->lr.team_cap, 12); ++i) fprintf(stderr, " %d", all[dh::LRT_LIST + i]);
    fprintf(stderr, "\n");
    for (int t = 0; t < b->lr_teams; ++t) {
      const int32_t* d = all.data() + dh::LRT_LIST + b->lr.team_cap + 8 * t;
      static int base = 0; if (t == 0) { int32_t v = 0; (void)hipMemcpy(&v, b->lr_team_state.p + dh::LRT_LIST + b->lr.team_cap + 8 * 4096, 4, hipMemcpyDeviceToHost); base = v; }
      if (d[0]) fprintf(stderr, "  team %d: junction %d | x10us after lr_kernel's block 0 started: entry %d, got it %d, set up +%d, R strips +%d, M strips +%d, winner +%d, end +%d = %d\n", t, d[1], d[2] - base, d[3] - base, d[4] - d[3], d[5] - d[4], d[6] - d[5], d[7] - d[6], d[0] - d[7], d[0] - base);
    }
  }
#endif
  out[0] = b->lr_teams;
  out[1] = std::min(v[dh::LRT_COUNT], b->lr.team_cap);   // junctions the teams took (the list's capacity bounds it)
  out[2] = v[dh::LRT_TAKEN];
  out[3] = v[dh::LRT_ERROR];
  return rc;
}

int dellyhip_batch_dp_kernel_ms(dellyhip_ctx* c, dellyhip_batch* b, double* ms_dp) {
  if (!b || !ms_dp) return fail(DELLYHIP_E_ARG, "null argument");
  (void)c;
  *ms_dp = b->ms_dp_last;
  return 0;
}

int dellyhip_batch_kernel_ms(dellyhip_ctx* c, dellyhip_batch* b, double* ms_split, double* ms_msa, int32_t* launches) {
  if (!b) return fail(DELLYHIP_E_ARG, "null argument");
  int rc = dellyhip_batch_sync(c, b);
  if (rc) return rc;
  int L = std::max(1, b->launches);
  if (ms_split) *ms_split = b->ms_split / L;
  if (ms_msa) *ms_msa = b->ms_msa / L;
  if (launches) *launches = b->launches;
  b->ms_dp_last = b->ms_dp / L;
  b->ms_split = b->ms_msa = b->ms_dp = 0;
  b->launches = 0;
  return 0;
}

// device-side compaction of a run batch: b->blob_off[i] = bytes of junctions < i, b->blob_compact = the used bytes
// back to back; *used = total.  The stream is synchronised on return.
static int compact_batch(dellyhip_ctx* c, dellyhip_batch* b, std::vector<uint64_t>& off, uint64_t* used) {
  int rc;
  *used = 0;
  off.assign((size_t)b->n + 1, 0);
  if (b->n == 0) return 0;
  if (b->fetch_pending && b->fetch_ev) HIPCHK(hipEventSynchronize(b->fetch_ev));   // (it uses the same offset / compaction buffers)
  if ((rc = b->blob_off.reserve(blob_off_words((size_t)b->n)))) return rc;
  launch_blob_offsets(c->stream, b->res.p, b->n, b->blob_off.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(off.data(), b->blob_off.p, off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *used = off[b->n];
  if (*used > 0) {
    if ((rc = b->blob_compact.reserve(*used))) return rc;
    hipLaunchKernelGGL(blob_gather_kernel, dim3(std::min(b->n, c->n_cu * 16)), dim3(dh::WAVE), 0, c->stream, b->res.p,
                       b->out_blob.p, b->blob_off.p, b->blob_compact.p, b->n, ~0ull);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

// blob offsets of a record after compaction: its three pieces start at `at`
static void rebase_offsets(dellyhip_result& R, uint64_t at) {
  const uint64_t l0 = (uint64_t)std::max(R.cons_len, 0), l1 = (uint64_t)std::max(R.allele_len, 0),
                 l2 = 2ull * (uint64_t)std::max(R.aln_len, 0);
  R.cons_off = l0 ? at : 0;
  at += l0;
  R.allele_off = l1 ? at : 0;
  at += l1;
  R.aln_off = l2 ? at : 0;
}

int dellyhip_batch_fetch(dellyhip_ctx* c, dellyhip_batch* b, dellyhip_result* results, char* out_blob,
                         uint64_t out_blob_cap, uint64_t* out_blob_len) {
  if (!c || !b || (!results && b->n)) return fail(DELLYHIP_E_ARG, "null argument");
  if (b->n && !b->ever_run) return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch: the batch has not been run");
  HIPCHK(hipSetDevice(c->device));
  int rc = dellyhip_batch_sync(c, b);
  if (rc) return rc;
  uint64_t used = 0;
  if (b->n) {
    // compact on the device, move only the bytes that are used (a few hundred per junction
    // instead of the fixed slot)
    std::vector<uint64_t> off;
    if ((rc = compact_batch(c, b, off, &used))) return rc;
    if (used > 0 && (!out_blob || used > out_blob_cap)) {
      if (out_blob_len) *out_blob_len = used;
      return fail(DELLYHIP_E_ARG, "out_blob too small");
    }
    HIPCHK(hipMemcpyAsync(results, b->res.p, b->n * sizeof(dellyhip_result), hipMemcpyDeviceToHost, c->stream));
    if (used > 0) HIPCHK(hipMemcpyAsync(out_blob, b->blob_compact.p, used, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < b->n; ++i) {
      rebase_offsets(results[i], off[i]);
#ifndef DH_LR_TIMING   // (the timing build of tools/ hands its phase times out in this field)
      results[i].reserved = 0;   // (transient kernel state -- SPS_DONE and the like -- does not cross the ABI)
#endif
    }
  }
  if (out_blob_len) *out_blob_len = used;
  return 0;
}

static hipStream_t device_download_stream(int device);   // (the device's download stream of the pipelined host path, below)

int dellyhip_batch_fetch_begin(dellyhip_ctx* c, dellyhip_batch* b, dellyhip_result* results, char* out_blob, uint64_t out_blob_cap) {
  if (!c || !b || (!results && b->n)) return fail(DELLYHIP_E_ARG, "null argument");
  if (b->lazy) return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch_begin: not for the slots of a dellyhip_stream");
  if (b->n && !b->ever_run) return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch_begin: the batch has not been run");
  if (b->fetch_pending) return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch_begin: a fetch of this batch is in flight (dellyhip_batch_fetch_end first)");
  HIPCHK(hipSetDevice(c->device));
  if (!b->fetch_status) {
    b->fetch_status = static_cast<uint64_t*>(PinPool::get().take(64, &b->fetch_status_bytes));
    if (!b->fetch_status) return fail(DELLYHIP_E_NOMEM, "hipHostMalloc (fetch status)");
  }
  b->fetch_status[0] = b->fetch_status[1] = 0;
  if (b->n == 0) { b->fetch_pending = true; return 0; }
  void *d_res = nullptr, *d_blob = nullptr, *d_status = nullptr;
  if (hipHostGetDevicePointer(&d_res, results, 0) != hipSuccess || (out_blob && out_blob_cap && hipHostGetDevicePointer(&d_blob, out_blob, 0) != hipSuccess)) {
    (void)hipGetLastError();
    return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch_begin: results / out_blob must be pinned host memory (dellyhip_host_register, hipHostMalloc): kernels write into it");
  }
  HIPCHK(hipHostGetDevicePointer(&d_status, b->fetch_status, 0));
  const uint64_t cap = d_blob ? out_blob_cap : 0;
  int rc;
  if ((rc = b->blob_off.reserve(blob_off_words((size_t)b->n)))) return rc;
  // (the used bytes are not known here: room for the smaller of everything the batch can hold and what the caller can take)
  // the kernels' cap is what was RESERVED here, not the caller's: a record with lengths beyond its out_stride share would
  // otherwise pass `off[n] <= cap` and overrun blob_compact.  (+16: the compacted bytes start at the destination's alignment mod 16.)
  const uint64_t room = std::max<uint64_t>(std::min<uint64_t>(cap, (uint64_t)b->n * b->out_stride), 16);
  if ((rc = b->blob_compact.reserve((size_t)room + 16))) return rc;
  const uint64_t kcap = std::min<uint64_t>(cap, room);
  uint8_t* compact = b->blob_compact.p + (reinterpret_cast<uintptr_t>(d_blob) & 15);
  if (!b->fetch_ev) HIPCHK(hipEventCreateWithFlags(&b->fetch_ev, hipEventDisableTiming));
  // on the device's download stream (the one dellyhip_stream returns its results on: verified to run beside the compute streams), behind
  // the end of the batch's run: the next launch on the batch's compute stream does not queue behind a PCIe-bound kernel.
  // DELLYHIP_FETCH_SAME_STREAM=1: on the stream of the run itself (A/B: 26.6 instead of 31.2 M junctions/s in bench.py's N > 1 step; the
  // high-priority upload stream instead of the low-priority download stream: no difference, tools/gpu_r05_u.sh)
  hipStream_t run_s = b->run_stream ? b->run_stream : c->stream;
  hipStream_t s = env_on("DELLYHIP_FETCH_SAME_STREAM") ? run_s : device_download_stream(c->device);
  if (!s) s = run_s;
  if (s != run_s && b->pending && b->last) HIPCHK(hipStreamWaitEvent(s, b->last, 0));
  launch_blob_offsets(s, b->res.p, b->n, b->blob_off.p);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(blob_gather_kernel, dim3(std::min(b->n, c->n_cu * 16)), dim3(dh::WAVE), 0, s, b->res.p, b->out_blob.p, b->blob_off.p,
                     compact, b->n, kcap);
  HIPCHK(hipGetLastError());
  const int32_t* lrt = (b->lr_teams > 0 && b->lr_team_state.p) ? b->lr_team_state.p + dh::LRT_ERROR : nullptr;
  hipLaunchKernelGGL(fetch_out_kernel, dim3(std::max(1, std::min(c->n_cu * 4, (b->n + 3) / 4))), dim3(256), 0, s, b->res.p, b->blob_off.p, b->n,
                     compact, static_cast<dellyhip_result*>(d_res), static_cast<uint8_t*>(d_blob), kcap,
                     static_cast<uint64_t*>(d_status), lrt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b->fetch_ev, s));
  b->fetch_stream = s;
  b->fetch_pending = true;
  return 0;
}

int dellyhip_batch_fetch_end(dellyhip_ctx* c, dellyhip_batch* b, uint64_t* out_blob_len) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "null argument");
  if (!b->fetch_pending) return fail(DELLYHIP_E_ARG, "dellyhip_batch_fetch_end: no fetch of this batch is in flight");
  HIPCHK(hipSetDevice(c->device));
  b->fetch_pending = false;
  if (out_blob_len) *out_blob_len = 0;
  if (b->n == 0) return 0;
  hipError_t e = hipEventSynchronize(b->fetch_ev);
  if (e != hipSuccess) return fail(DELLYHIP_E_RUNTIME, "dellyhip_batch_fetch_end: kernel execution", e);
  const uint64_t used = b->fetch_status[0], flags = b->fetch_status[1];
  if (out_blob_len) *out_blob_len = used;
  if (flags & 2) return fail(DELLYHIP_E_RUNTIME, "a long-read team of wavefronts gave up waiting: the junctions it was to sweep are not refined");
  if (flags & 1) return fail(DELLYHIP_E_ARG, "out_blob too small");
  return 0;
}

#ifdef DH_LR_TIMING
// profiling builds only (tools/lrc_phases.py): the phase clocks of lrmsa_kernel / lrwfa_kernel (lrmsa_kernel.hpp), read and cleared
extern "C" int dellyhip_debug_lrt(uint64_t* out, int n) {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dh::dh_lrt), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 32; ++i) out[i] = h[i];
  unsigned long long pr[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(pr, HIP_SYMBOL(dh::dh_lrt_pairs), sizeof pr) == hipSuccess && n > 21) { out[20] = pr[0]; out[21] = pr[1]; }
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dh::dh_lrt_pairs), h, sizeof pr);
  return hipMemcpyToSymbol(HIP_SYMBOL(dh::dh_lrt), h, sizeof h) == hipSuccess ? 0 : -1;
}
#endif

#ifdef DH_SPS_DBG
// profiling builds only (tools/sps_dbg.py): the counters of sparse_needle.hpp, read and cleared
extern "C" int dellyhip_debug_read(uint64_t* out, int n) {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dh::dh_dbg), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 16; ++i) out[i] = h[i];
  memset(h, 0, sizeof h);
  return hipMemcpyToSymbol(HIP_SYMBOL(dh::dh_dbg), h, sizeof h) == hipSuccess ? 0 : -1;
}
#endif

// ---- multi-GPU: cost-balanced sharding + RCCL gather of the results ---------------------------
struct dellyhip_comm {
  int32_t rank = 0, world = 1;
  std::unique_ptr<dh::Link> link;      // null: one rank, no transport
  DevBuf<uint8_t> rec_all, blob_all;   // root: receive areas
};

int dellyhip_shard_by_cost(const dellyhip_params* P, int32_t n, const dellyhip_junction* junc, const uint64_t* seq_off,
                           uint64_t n_seq, int32_t world, int32_t* owner) {
  if (n < 0 || world < 1 || (n && (!junc || !seq_off || !owner))) return fail(DELLYHIP_E_ARG, "bad argument");
  std::vector<double> cost((size_t)n);
  for (int i = 0; i < n; ++i) {
    const dellyhip_junction& J = junc[i];
    if (J.n_seq < 0 || J.seq_first > n_seq || J.seq_first + (uint64_t)J.n_seq > n_seq) return fail(DELLYHIP_E_ARG, "junction sequence range");
    if (seq_off[J.seq_first + J.n_seq] < seq_off[J.seq_first]) return fail(DELLYHIP_E_ARG, "seq_off is not monotonic");
    const double N = (double)J.n_seq;
    const double bytes = (double)(seq_off[J.seq_first + J.n_seq] - seq_off[J.seq_first]);
    const double L = N > 0 ? bytes / N : 0.0;
    const double m = L;   // consensus ~ one read length (given consensus: exactly)
    const double span = std::max(0.0, (double)J.sv_end - (double)J.sv_start);
    const double win = (J.svt == 2 && P && span <= (double)P->indelsize) ? 2.0 * m + span : 4.0 * m;   // src/split.h:116-117
    cost[i] = (N > 1 ? N * N * L * L / 2.0 + (N - 1) * L * L * 4.0 : 0.0) + 2.0 * m * win + 1.0;
  }
  dh::balance_by_cost(cost.data(), n, world, owner);
  return 0;
}

int dellyhip_comm_unique_id(void* id128) {
  if (!id128) return fail(DELLYHIP_E_ARG, "null argument");
  dh::RcclApi& A = dh::rccl_api();
  if (!A.error.empty()) return fail(DELLYHIP_E_RUNTIME, A.error.c_str());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
  ncclUniqueId id;
  const ncclResult_t r = A.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(DELLYHIP_E_RUNTIME, A.GetErrorString(r));
  memcpy(id128, &id, sizeof id);
  return 0;
}

int dellyhip_comm_create(dellyhip_ctx* c, const void* id128, int32_t rank, int32_t world, dellyhip_comm** out) {
  if (!c || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<dellyhip_comm> m(new dellyhip_comm());
  m->rank = rank;
  m->world = world;
  if (world > 1 || id128) {   // (world == 1 with an id: a real one-rank RCCL communicator -- exercises the RCCL path on one GPU)
    std::unique_ptr<dh::RcclLink> L(new dh::RcclLink());
    if (int rc = L->init(id128, rank, world)) return fail(rc, L->err.c_str());
    m->link = std::move(L);
  }
  *out = m.release();
  return 0;
}

int dellyhip_comm_create_hostlink(dellyhip_ctx* c, const char* name, int32_t rank, int32_t world, dellyhip_comm** out) {
  if (!out || !name || world < 1 || rank < 0 || rank >= world) return fail(DELLYHIP_E_ARG, "bad argument");
  if (c) HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<dellyhip_comm> m(new dellyhip_comm());
  m->rank = rank;
  m->world = world;
  std::unique_ptr<dh::HostLink> L(new dh::HostLink());
  if (int rc = L->init(name, rank, world, c != nullptr)) return fail(rc, L->err.c_str());
  m->link = std::move(L);
  *out = m.release();
  return 0;
}

int dellyhip_comm_info(dellyhip_comm* m, int32_t* rank, int32_t* world, int32_t* transport_ranks, char* kind16) {
  if (!m) return fail(DELLYHIP_E_ARG, "null argument");
  if (rank) *rank = m->rank;
  if (world) *world = m->world;
  if (transport_ranks) *transport_ranks = m->link ? m->link->transport_ranks() : 1;
  if (kind16) snprintf(kind16, 16, "%s", m->link ? m->link->kind() : "none");
  return 0;
}

void dellyhip_comm_destroy(dellyhip_comm* m) {
  if (!m) return;
  delete m;
}

// A local failure must not leave the other ranks inside a collective: every rank ALWAYS takes part in the exchange of the
// (count, bytes) pairs -- a rank whose batch could not be synchronised or compacted sends count = GATHER_ERR -- and in a second
// one-word exchange after the root has sized its receive areas; all ranks then agree on whether the Send / Recv group runs.
static const uint64_t GATHER_ERR = ~0ull;

// first exchange: all[2 r] = records of rank r, all[2 r + 1] = bytes of its compact blob.  Every rank returns the same
// verdict: 0, or an error if ANY rank reported a failure (the failing rank keeps its own message).
static int exchange_sizes(dellyhip_comm* m, hipStream_t s, uint64_t count, uint64_t bytes, int local_rc, const std::string& local_err,
                          std::vector<uint64_t>& all) {
  const int W = m->world;
  all.assign(2 * (size_t)W, 0);
  if (!m->link) {
    if (local_rc) return fail(local_rc, local_err.c_str());
    all[0] = count;
    all[1] = bytes;
    return 0;
  }
  const uint64_t mine[2] = {local_rc ? GATHER_ERR : count, local_rc ? 0 : bytes};
  if (int rc = m->link->allgather2(mine, all.data(), s)) return fail(rc, m->link->err.c_str());
  for (int r2 = 0; r2 < W; ++r2)
    if (all[2 * r2] == GATHER_ERR) {
      if (local_rc) return fail(local_rc, local_err.c_str());
      char msg[96];
      snprintf(msg, sizeof msg, "dellyhip_gather_results: rank %d failed before the exchange", r2);
      return fail(DELLYHIP_E_RUNTIME, msg);
    }
  return 0;
}

// second exchange: is the root ready to receive?  (one word per rank; only the root's matters)
static int exchange_ready(dellyhip_comm* m, hipStream_t s, int32_t root, bool root_failed) {
  if (!m->link) return root_failed ? fail(DELLYHIP_E_NOMEM, "dellyhip_gather_results: the root could not allocate its receive areas") : 0;
  const int W = m->world;
  const uint64_t ready[2] = {(m->rank == root && root_failed) ? GATHER_ERR : 0, 0};
  std::vector<uint64_t> seen(2 * (size_t)W, 0);
  if (int rc = m->link->allgather2(ready, seen.data(), s)) return fail(rc, m->link->err.c_str());
  if (seen[2 * (size_t)root] == GATHER_ERR)
    return fail(DELLYHIP_E_NOMEM, "dellyhip_gather_results: the root could not allocate its receive areas");
  return 0;
}

int dellyhip_comm_exchange_sizes(dellyhip_ctx* c, dellyhip_comm* m, uint64_t count, uint64_t bytes, int32_t failed, uint64_t* all) {
  if (!m || !all) return fail(DELLYHIP_E_ARG, "null argument");
  if (c) HIPCHK(hipSetDevice(c->device));
  else if (m->link && std::string(m->link->kind()) == "rccl") return fail(DELLYHIP_E_ARG, "an RCCL communicator needs its context");
  std::vector<uint64_t> v;
  const int rc = exchange_sizes(m, c ? c->stream : nullptr, count, bytes, failed ? DELLYHIP_E_RUNTIME : 0,
                                "dellyhip_comm_exchange_sizes: this rank reported a failure", v);
  for (size_t i = 0; i < v.size(); ++i) all[i] = v[i];
  return rc;
}

int dellyhip_comm_exchange_ready(dellyhip_ctx* c, dellyhip_comm* m, int32_t root, int32_t root_failed) {
  if (!m || root < 0 || root >= m->world) return fail(DELLYHIP_E_ARG, "bad argument");
  if (c) HIPCHK(hipSetDevice(c->device));
  else if (m->link && std::string(m->link->kind()) == "rccl") return fail(DELLYHIP_E_ARG, "an RCCL communicator needs its context");
  return exchange_ready(m, c ? c->stream : nullptr, root, root_failed != 0);
}

// gatherv of one opaque payload per rank: the protocol of gather_device (sizes, root readiness, one send / receive group)
// on caller buffers -- device pointers with a context, host pointers on a device-less hostlink
int dellyhip_comm_gather_bytes(dellyhip_ctx* c, dellyhip_comm* m, int32_t root, const void* mine, uint64_t bytes, void* out,
                               uint64_t out_cap, uint64_t* sizes) {
  if (!m || root < 0 || root >= m->world || (bytes && !mine)) return fail(DELLYHIP_E_ARG, "bad argument");
  if (c) HIPCHK(hipSetDevice(c->device));
  else if (m->link && std::string(m->link->kind()) == "rccl") return fail(DELLYHIP_E_ARG, "an RCCL communicator needs its context");
  hipStream_t s = c ? c->stream : nullptr;
  const int W = m->world;
  const bool is_root = m->rank == root;
  std::vector<uint64_t> all;
  if (int rc = exchange_sizes(m, s, 1, bytes, 0, std::string(), all)) return rc;
  uint64_t tot = 0;
  std::vector<uint64_t> first(W + 1, 0);
  for (int r = 0; r < W; ++r) { first[r] = tot; tot += all[2 * r + 1]; }
  if (sizes)
    for (int r = 0; r < W; ++r) sizes[r] = all[2 * r + 1];
  const bool short_buf = is_root && (tot > out_cap || (tot && !out));
  if (int rc = exchange_ready(m, s, root, short_buf)) return rc;
  uint8_t* o = static_cast<uint8_t*>(out);
  if (W > 1) {
    dh::Link& L = *m->link;
    if (int rc = L.group_begin()) return fail(rc, L.err.c_str());
    if (!is_root) L.send(mine, bytes, root, s);
    else
      for (int q = 0; q < W; ++q)
        if (q != root) L.recv(o + first[q], all[2 * q + 1], q, s);
    if (int rc = L.group_end(s)) return fail(rc, L.err.c_str());
  }
  if (is_root && bytes) {
    if (c) { HIPCHK(hipMemcpyAsync(o + first[root], mine, bytes, hipMemcpyDefault, s)); HIPCHK(hipStreamSynchronize(s)); }
    else memcpy(o + first[root], mine, bytes);
  }
  return 0;
}

// the exchange itself: afterwards the root holds every rank's records (rank order) and compact blobs in HBM
static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int gather_device(dellyhip_ctx* c, dellyhip_comm* m, dellyhip_batch* b, int32_t root, std::vector<uint64_t>& all,
                         std::vector<uint64_t>& first_n, std::vector<uint64_t>& first_b, const void** d_rec, const void** d_blob) {
  std::vector<uint64_t> off;
  uint64_t used = 0;
  static const bool trace = getenv("DELLYHIP_TRACE_GATHER") != nullptr;   // (debugging aid: where a gather's host time goes)
  const double tg0 = trace ? now_s() : 0;
  int local_rc = dellyhip_batch_sync(c, b);
  const double tg1 = trace ? now_s() : 0;
  if (!local_rc) local_rc = compact_batch(c, b, off, &used);
  if (trace) fprintf(stderr, "gather: wait for the batch %.3f ms, compaction %.3f ms\n", (tg1 - tg0) * 1e3, (now_s() - tg1) * 1e3);
  if (!local_rc && getenv("DELLYHIP_TEST_FAIL_GATHER_RANK") && atoi(getenv("DELLYHIP_TEST_FAIL_GATHER_RANK")) == m->rank)
    local_rc = fail(DELLYHIP_E_RUNTIME, "dellyhip_gather_results: failure injected by DELLYHIP_TEST_FAIL_GATHER_RANK");   // (tests of the abort protocol)
  const std::string local_err = local_rc ? g_err : std::string();
  const int W = m->world;
  const bool is_root = m->rank == root;
  const double tg2 = trace ? now_s() : 0;
  if (int rc = exchange_sizes(m, c->stream, (uint64_t)b->n, used, local_rc, local_err, all)) return rc;
  if (trace) fprintf(stderr, "gather: size exchange %.3f ms\n", (now_s() - tg2) * 1e3);
  uint64_t tot_n = 0, tot_b = 0;
  first_n.assign(W + 1, 0);
  first_b.assign(W + 1, 0);
  for (int r = 0; r < W; ++r) {
    first_n[r] = tot_n;
    first_b[r] = tot_b;
    tot_n += all[2 * r];
    tot_b += all[2 * r + 1];
  }
  first_n[W] = tot_n;
  first_b[W] = tot_b;
  const uint8_t* rec_src = reinterpret_cast<const uint8_t*>(b->res.p);
  *d_rec = rec_src;
  *d_blob = b->blob_compact.p;
  if (W > 1) {
    dh::Link& L = *m->link;
    int root_rc = 0;
    if (is_root) {
      if ((root_rc = m->rec_all.reserve(std::max<uint64_t>(tot_n * sizeof(dellyhip_result), 1)))) root_rc = DELLYHIP_E_NOMEM;
      else if ((root_rc = m->blob_all.reserve(std::max<uint64_t>(tot_b, 1)))) root_rc = DELLYHIP_E_NOMEM;
      if (!root_rc && getenv("DELLYHIP_TEST_FAIL_GATHER_ROOT")) root_rc = DELLYHIP_E_NOMEM;   // (tests of the abort protocol)
    }
    if (int rc = exchange_ready(m, c->stream, root, is_root && root_rc)) return rc;
    if (int rc = L.group_begin()) return fail(rc, L.err.c_str());
    if (!is_root) {
      if (b->n) L.send(rec_src, (uint64_t)b->n * sizeof(dellyhip_result), root, c->stream);
      if (used) L.send(b->blob_compact.p, used, root, c->stream);
    } else {
      for (int q = 0; q < W; ++q) {
        if (q == root) continue;
        if (all[2 * q]) L.recv(m->rec_all.p + first_n[q] * sizeof(dellyhip_result), all[2 * q] * sizeof(dellyhip_result), q, c->stream);
        if (all[2 * q + 1]) L.recv(m->blob_all.p + first_b[q], all[2 * q + 1], q, c->stream);
      }
    }
    if (int rc = L.group_end(c->stream)) return fail(rc, L.err.c_str());
    if (is_root) {   // the root's own share, device to device
      if (b->n) HIPCHK(hipMemcpyAsync(m->rec_all.p + first_n[root] * sizeof(dellyhip_result), rec_src, (size_t)b->n * sizeof(dellyhip_result), hipMemcpyDeviceToDevice, c->stream));
      if (used) HIPCHK(hipMemcpyAsync(m->blob_all.p + first_b[root], b->blob_compact.p, used, hipMemcpyDeviceToDevice, c->stream));
      *d_rec = m->rec_all.p;
      *d_blob = m->blob_all.p;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

int dellyhip_gather_results_device(dellyhip_ctx* c, dellyhip_comm* m, dellyhip_batch* b, int32_t root, const void** d_records,
                                   uint64_t* n_results, const void** d_blob, uint64_t* blob_bytes) {
  if (!c || !m || !b || root < 0 || root >= m->world) return fail(DELLYHIP_E_ARG, "bad argument");
  if (b->n && !b->ever_run) return fail(DELLYHIP_E_ARG, "dellyhip_gather_results_device: the batch has not been run");
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint64_t> all, first_n, first_b;
  const void *dr = nullptr, *db = nullptr;
  int rc = gather_device(c, m, b, root, all, first_n, first_b, &dr, &db);
  if (rc) return rc;
  const bool is_root = m->rank == root;
  if (d_records) *d_records = is_root ? dr : nullptr;
  if (d_blob) *d_blob = is_root ? db : nullptr;
  if (n_results) *n_results = first_n[m->world];
  if (blob_bytes) *blob_bytes = first_b[m->world];
  return 0;
}

int dellyhip_rebase_gathered(dellyhip_result* results, uint64_t n_results, int32_t world, const uint64_t* counts, const uint64_t* bytes) {
  if (world < 1 || !counts || !bytes || (n_results && !results)) return fail(DELLYHIP_E_ARG, "bad argument");
  uint64_t k = 0, base = 0;
  for (int r = 0; r < world; ++r) {
    uint64_t at = base;
    for (uint64_t q = 0; q < counts[r]; ++q, ++k) {
      if (k >= n_results) return fail(DELLYHIP_E_ARG, "dellyhip_rebase_gathered: counts exceed the records");
      dellyhip_result& R = results[k];
      const uint64_t len = (uint64_t)std::max(R.cons_len, 0) + (uint64_t)std::max(R.allele_len, 0) + 2ull * (uint64_t)std::max(R.aln_len, 0);
      rebase_offsets(R, at);
      R.reserved = 0;
      at += len;
    }
    if (at - base != bytes[r]) return fail(DELLYHIP_E_RUNTIME, "dellyhip_rebase_gathered: a rank's records do not add up to its blob bytes");
    base += bytes[r];
  }
  if (k != n_results) return fail(DELLYHIP_E_ARG, "dellyhip_rebase_gathered: counts do not cover the records");
  return 0;
}

int dellyhip_gather_results(dellyhip_ctx* c, dellyhip_comm* m, dellyhip_batch* b, int32_t root, dellyhip_result* results,
                            uint64_t results_cap, uint64_t* n_results, char* out_blob, uint64_t out_blob_cap,
                            uint64_t* out_blob_len, int32_t* counts) {
  if (!c || !m || !b || root < 0 || root >= m->world) return fail(DELLYHIP_E_ARG, "bad argument");
  if (b->n && !b->ever_run) return fail(DELLYHIP_E_ARG, "dellyhip_gather_results: the batch has not been run");
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint64_t> all, first_n, first_b;
  const void *dr = nullptr, *db = nullptr;
  // (a root whose buffers turn out too small has still taken part in the exchange: the senders never block on it)
  int rc = gather_device(c, m, b, root, all, first_n, first_b, &dr, &db);
  if (rc) return rc;
  const int W = m->world;
  const uint64_t tot_n = first_n[W], tot_b = first_b[W];
  if (n_results) *n_results = tot_n;
  if (out_blob_len) *out_blob_len = tot_b;
  if (m->rank != root) return 0;
  if (!((tot_n == 0 || (results && tot_n <= results_cap)) && (tot_b == 0 || (out_blob && tot_b <= out_blob_cap))))
    return fail(DELLYHIP_E_ARG, "dellyhip_gather_results: results / out_blob too small (needed sizes returned)");
  if (counts)
    for (int r = 0; r < W; ++r) counts[r] = (int32_t)all[2 * r];
  if (tot_n) {
    const bool trace = getenv("DELLYHIP_TRACE_GATHER") != nullptr;
    const double t0 = trace ? now_s() : 0;
    HIPCHK(hipMemcpyAsync(results, dr, tot_n * sizeof(dellyhip_result), hipMemcpyDeviceToHost, c->stream));
    if (tot_b) HIPCHK(hipMemcpyAsync(out_blob, db, tot_b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (trace) fprintf(stderr, "gather: download of %llu + %llu bytes %.3f ms\n", (unsigned long long)(tot_n * sizeof(dellyhip_result)), (unsigned long long)tot_b, (now_s() - t0) * 1e3);
    // blob offsets: each rank's pieces lie back to back in its compact blob in junction order
    std::vector<uint64_t> cnt(W), byt(W);
    for (int r = 0; r < W; ++r) { cnt[r] = first_n[r + 1] - first_n[r]; byt[r] = first_b[r + 1] - first_b[r]; }
    const double t1 = trace ? now_s() : 0;
    if (dellyhip_rebase_gathered(results, tot_n, W, cnt.data(), byt.data())) return DELLYHIP_E_RUNTIME;
    if (trace) fprintf(stderr, "gather: rebase on the host %.3f ms\n", (now_s() - t1) * 1e3);
  }
  return 0;
}

// ---- pipelined host-buffer path: dellyhip_stream (SURVEY.md 8d "GPU time includes H2D/D2H and host marshalling") ----
// A stream owns `depth` slots.  Each slot is a child context (own HIP stream, scratch area and work counters, sharing the
// parent's resident chromosomes), a recycled batch object, a pinned staging arena for the inputs and a pinned block for
// the outputs.  submit() validates and routes the junctions, copies the inputs into the arena and enqueues -- one H2D
// copy, the kernels, the device-side compaction, the D2H copies -- without waiting; collect() waits for the oldest slot
// and hands out pointers into its pinned output block.  Nothing is allocated per batch once the buffers have grown to
// the batch size; with depth >= 2 the copies of one slot overlap the kernels of the others.
struct StreamHeader {      // head of the pinned output block
  uint64_t used;           // bytes of the compact blob
  int32_t sps_left;        // junctions split_sparse_kernel left to the dense kernels
  int32_t pad_;
};
struct StreamSlot {
  dellyhip_ctx* ctx = nullptr;
  dellyhip_batch* b = nullptr;
  Arena in;
  PinBuf<uint8_t> out;
  DevBuf<uint8_t> d_rec;   // StreamHeader (64 bytes) | rebased records: the image of the head of the pinned block
  hipEvent_t done = nullptr, up_done = nullptr, comp_done = nullptr;
  int state = 0;           // 0 free, 1 submitted, 2 collected (the caller still reads its output block)
  bool down_pending = false;   // compaction enqueued, D2H copies not yet (slot_pump enqueues them once comp_done has fired)
  int32_t n = 0;
  uint64_t tag = 0;
  uint64_t blob_copied = 0, blob_cap = 0;
  size_t o_rec = 0, o_len = 0, o_blob = 0;
};
// ---- do two HIP streams run concurrently?  (they do not when the runtime mapped them onto one hardware queue) --------------
__global__ void probe_wait_kernel(int* flag, int* out, long long max_ticks) {
  const long long t0 = wall_clock64();   // 100 MHz
  int seen = 0;
  while (!seen && wall_clock64() - t0 < max_ticks) seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *out = seen;
}
__global__ void probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a waits (at most 0.5 ms) for a flag that only a kernel on b sets: true = b's kernel ran while a's was running
static bool streams_run_concurrently(hipStream_t a, hipStream_t b, int* scratch2) {
  if (hipMemsetAsync(scratch2, 0, 2 * sizeof(int), a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, a, scratch2, scratch2 + 1, 50000ll);
  hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, b, scratch2);
  int seen = 0;
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
      hipMemcpy(&seen, scratch2 + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return seen != 0;
}

struct DeviceStreams {
  std::mutex mu;
  bool made = false;
  hipStream_t up = nullptr, down = nullptr, comp[2] = {nullptr, nullptr};
  int probed_pairs = 0;   // candidate streams created until two ran concurrently
};
static DeviceStreams& device_streams(int device) {
  static std::mutex mu;
  static std::map<int, std::unique_ptr<DeviceStreams>> all;
  std::lock_guard<std::mutex> g(mu);
  auto& p = all[device];
  if (!p) p.reset(new DeviceStreams());
  return *p;
}
struct dellyhip_stream {
  dellyhip_ctx* parent = nullptr;
  int with_msa = 0, want_alignment = 0;
  bool zero_copy = false;   // dellyhip_stream_zero_copy: pinned sequence bytes are read in place
  std::vector<StreamSlot> slots;
  uint64_t n_submit = 0, n_collect = 0;
  // The copies run on two streams of their own, created with priorities other than the compute streams': the runtime keeps
  // one pool of hardware queues per priority, so a copy never sits in the hardware queue of another slot's kernels (streams
  // of one priority share at most GPU_MAX_HW_QUEUES = 4 queues; commands of one hardware queue execute in order, and a
  // slot's 0.2 ms download in front of the next slot's kernels was measured to cost 40 % of the throughput).
  hipStream_t s_up = nullptr, s_down = nullptr;
  // ... and the slots' kernels on TWO compute streams, even and odd slots alternating: consecutive batches may overlap
  // (the tail of one sparse kernel -- 2.4 wavefronts per slot at 10 000 junctions -- under the head of the next) without
  // one stream per slot competing for the four hardware queues of the normal priority.
  hipStream_t s_comp[2] = {nullptr, nullptr};
  // (The compaction kernels of a slot stay on its compute stream.  A third stream for them was tried in round 3 --
  // so that the sparse kernel of the slot after next need not wait for them -- and lost: 24 M junctions/s against 29.5,
  // tools/stream_matrix.sh.)
  int held = -1;                    // slot whose output the caller holds since the last collect()
  int test_fail_collect = 0;        // tests: the next n collects fail after their batch has finished (env DELLYHIP_TEST_FAIL_COLLECT)
  int test_fail_submit = 0;         // tests: the next n submits fail after their upload has been enqueued (env DELLYHIP_TEST_FAIL_SUBMIT)
  double blob_per_junction = 0;     // running estimate: bytes of compact blob per junction (sizes the first D2H copy)
  // host seconds since creation / the last dellyhip_stream_stats(reset): validation + routing + staging | kernel launches |
  // compaction + download enqueue | waiting in collect | slow-path batches (leftovers routed at collect) | blob top-ups
  double t_stage = 0, t_launch = 0, t_down = 0, t_wait = 0;
  uint64_t n_slow = 0, n_topup = 0;
  // DELLYHIP_LOG=1: a line per collected batch on stderr in the reference's "[timestamp] stage" style (src/shortpe.h:76-77):
  // junctions, junctions/s since the stream was created, bytes up / down
  int log = 0;
  double t_created = 0;
  uint64_t log_junctions = 0, log_up = 0, log_down = 0;
};

namespace {

int slot_download(dellyhip_stream* st, StreamSlot& S, bool all_blob, hipStream_t s);

// offsets + gather + rebased records of slot S on its stream, then the D2H copies into the pinned block
int slot_compact_and_download(dellyhip_stream* st, StreamSlot& S, bool all_blob) {
  dellyhip_ctx* c = S.ctx;
  dellyhip_batch* b = S.b;
  const int n = b->n;
  int rc;
  if ((rc = b->blob_off.reserve_grow(blob_off_words((size_t)n))) || (rc = b->blob_compact.reserve_grow(std::max<uint64_t>((uint64_t)n * b->out_stride, 1))) ||
      (rc = S.d_rec.reserve_grow(64 + (size_t)std::max(n, 1) * sizeof(dellyhip_result))))
    return rc;
  hipStream_t s = c->stream;
  const bool split = st->s_down && S.comp_done && !all_blob;   // (the slow path stays on the slot's own stream)
  launch_blob_offsets(s, b->res.p, n, b->blob_off.p);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(blob_gather_records_kernel, dim3(std::min(n, c->n_cu * 16)), dim3(dh::WAVE), 0, s, b->res.p, b->out_blob.p,
                     b->blob_off.p, b->blob_compact.p, reinterpret_cast<dellyhip_result*>(S.d_rec.p + 64), n,
                     reinterpret_cast<uint64_t*>(S.d_rec.p), reinterpret_cast<int32_t*>(S.d_rec.p + 8),
                     c->counters.p ? c->counters.p + 31 : nullptr);
  HIPCHK(hipGetLastError());
  if (split) {
    HIPCHK(hipEventRecord(S.comp_done, s));
    // The copies are NOT enqueued here: the runtime hands a copy and the signal it waits for to an SDMA ring, a waiting
    // copy stalls every copy behind it in that ring -- the uploads of the following slots included (traced: each H2D
    // started 10-16 us after the previous slot's D2H had finished, so one sparse kernel ran at a time whatever the depth).
    // slot_pump() enqueues them once comp_done has fired: nothing in the ring ever waits.
    static const bool eager = getenv("DELLYHIP_STREAM_EAGER_DOWN") != nullptr;   // (measurement knob: the round-3a behaviour)
    if (eager) {
      HIPCHK(hipStreamWaitEvent(st->s_down, S.comp_done, 0));
      return slot_download(st, S, false, st->s_down);
    }
    S.down_pending = true;
    return 0;
  }
  return slot_download(st, S, all_blob, s);
}

// the D2H copies of slot S into its pinned block
int slot_download(dellyhip_stream* st, StreamSlot& S, bool all_blob, hipStream_t s) {
  dellyhip_batch* b = S.b;
  const int n = b->n;
  StreamHeader* H = reinterpret_cast<StreamHeader*>(S.out.p);
  S.down_pending = false;
  // header (bytes of compact blob, junctions the sparse kernel left) and records in ONE copy: small copies are blit kernels,
  // and a blit kernel on this stream waits for room on the chip behind the persistent wavefronts of the sparse kernels --
  // two of them per batch made this stream the bottleneck of the pipeline (0.6 ms per batch, profiles/r03/README.md)
  static_assert(sizeof(StreamHeader) <= 64 && offsetof(StreamHeader, sps_left) == 8, "StreamHeader layout");
  HIPCHK(hipMemcpyAsync(S.out.p, S.d_rec.p, 64 + (size_t)n * sizeof(dellyhip_result), hipMemcpyDeviceToHost, s));
  // the blob's size is only known on the device: copy what the previous batches predict (all of it after a slow path)
  uint64_t want = all_blob ? S.blob_cap : (uint64_t)(st->blob_per_junction * 1.06 * n) + 4096;
  want = std::min<uint64_t>(std::min<uint64_t>(want, S.blob_cap), (uint64_t)n * b->out_stride);
  if (all_blob) want = std::min<uint64_t>(want, H->used);   // (slow path: the header has been read already)
  S.blob_copied = want;
  if (want) HIPCHK(hipMemcpyAsync(S.out.p + S.o_blob, b->blob_compact.p, want, hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(S.done, s));
  return 0;
}

// enqueues the downloads of every submitted slot whose compaction has finished (oldest first); `must` = this slot's
// download is needed now: wait for its compaction
int slot_pump(dellyhip_stream* st, StreamSlot* must) {
  for (uint64_t k = st->n_collect; k < st->n_submit; ++k) {
    StreamSlot& S = st->slots[k % st->slots.size()];
    if (!S.down_pending) continue;
    if (&S == must) HIPCHK(hipEventSynchronize(S.comp_done));
    else if (hipEventQuery(S.comp_done) != hipSuccess) {
      (void)hipGetLastError();   // (hipErrorNotReady is not an error)
      continue;
    }
    int rc = slot_download(st, S, false, st->s_down);
    if (rc) return rc;
  }
  return 0;
}

// the compact blob turned out larger than the pinned block: a larger block (header, records and lengths are copied over)
int grow_out_block(StreamSlot& S, uint64_t blob_bytes) {
  PinBuf<uint8_t> bigger;
  int rc = bigger.reserve(S.o_blob + blob_bytes + 64);
  if (rc) return rc;
  memcpy(bigger.p, S.out.p, S.o_blob);
  std::swap(bigger.p, S.out.p);
  std::swap(bigger.n, S.out.n);
  std::swap(bigger.bytes_, S.out.bytes_);
  S.blob_cap = S.out.n - S.o_blob - 64;
  if (S.b) S.b->pin_cons_len = reinterpret_cast<int32_t*>(S.out.p + S.o_len);
  return 0;
}

// a lazy run whose sparse kernel left junctions behind: route them on the host, run the dense kernels (synchronous)
int finish_lazy(StreamSlot& S) {
  dellyhip_ctx* c = S.ctx;
  dellyhip_batch* b = S.b;
  int rc;
  if (b->with_msa == 1) {
    if (b->pin_cons_len) memcpy(b->h_cons_len.data(), b->pin_cons_len, (size_t)b->n * sizeof(int32_t));
    b->lazy = 0;
    if ((rc = route_after_msa(c, b))) return rc;   // (launch_early_sparse has set early_done: the sparse kernel is not launched again)
  } else {
    b->work.release();
    if ((rc = build_bins(b, c->params, BINS_LEFTOVER))) return rc;
    b->early_done = true;   // slots 30 / 31 of the counters belong to the sparse kernel that already ran
  }
  if ((rc = run_split(c, b, c->stream, false))) return rc;
  b->lazy_pending = false;
  return 0;
}

}  // namespace

// The four HIP streams of the pipelined path (upload, download, two compute streams) of a device: created on first use
static DeviceStreams& ensure_device_streams(int device) {
  DeviceStreams& D = device_streams(device);
  std::lock_guard<std::mutex> g(D.mu);
  // The four HIP streams of the pipelined path exist once per device and process and are never destroyed: the runtime
  // maps HIP streams onto a few hardware queues, commands of one hardware queue execute in order, and the mapping a
  // stream gets depends on the streams created and destroyed before it.
  if (!D.made) {
    D.made = true;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    // Two compute streams that really run side by side: with more HIP streams alive than hardware queues
    // (GPU_MAX_HW_QUEUES, 4 by default -- a torch.cuda.Stream() alone creates a pool of 32) two consecutively created
    // streams can share a queue, and the sparse kernels of consecutive slots then run one after the other: 19.4 instead of
    // 32.5 M junctions/s (tools/stream_matrix.sh).  Candidates are probed pairwise; the spare ones stay allocated
    // (destroying them would shift the mapping of streams created later).
    {
      hipStream_t cand[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      int ncand = 0;
      for (; ncand < 2; ++ncand)
        if (hipStreamCreateWithFlags(&cand[ncand], hipStreamNonBlocking) != hipSuccess) { cand[ncand] = nullptr; break; }
      int* probe = nullptr;
      int a = 0, b = 1;
      if (ncand == 2 && !getenv("DELLYHIP_STREAM_NO_PROBE") && dh::dev_alloc((void**)&probe, 2 * sizeof(int)) == hipSuccess) {
        bool ok = streams_run_concurrently(cand[0], cand[1], probe);
        while (!ok && ncand < 8) {
          if (hipStreamCreateWithFlags(&cand[ncand], hipStreamNonBlocking) != hipSuccess) { cand[ncand] = nullptr; break; }
          ++ncand;
          for (int i = 0; i + 1 < ncand && !ok; ++i)
            if (streams_run_concurrently(cand[i], cand[ncand - 1], probe)) { ok = true; a = i; b = ncand - 1; }
        }
        if (!ok) { a = 0; b = 1; }
        (void)hipDeviceSynchronize();
        dh::dev_free(probe);
      }
      D.comp[0] = cand[a];
      D.comp[1] = ncand >= 2 ? cand[b] : nullptr;
      D.probed_pairs = ncand;
    }
    if (least != greatest) {   // (numerically lower = higher priority; normal = 0 lies between)
      if (hipStreamCreateWithPriority(&D.up, hipStreamNonBlocking, greatest) != hipSuccess) D.up = nullptr;
      if (hipStreamCreateWithPriority(&D.down, hipStreamNonBlocking, least) != hipSuccess) D.down = nullptr;
    }
  }
  return D;
}

static hipStream_t device_download_stream(int device) { return ensure_device_streams(device).down; }

int dellyhip_compute_streams(dellyhip_ctx* c, void* out[2]) {
  if (!c || !out) return fail(DELLYHIP_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  DeviceStreams& D = ensure_device_streams(c->device);
  out[0] = D.comp[0];
  out[1] = D.comp[1] ? D.comp[1] : D.comp[0];
  return out[0] ? 0 : fail(DELLYHIP_E_RUNTIME, "no compute stream");
}

int dellyhip_stream_create(dellyhip_ctx* c, int32_t depth, int32_t with_msa, int32_t want_alignment, dellyhip_stream** out) {
  if (!c || !out || depth < 1 || depth > 8 || with_msa < 0 || with_msa > 2) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<dellyhip_stream> st(new dellyhip_stream());
  st->parent = c;
  st->with_msa = with_msa;
  st->want_alignment = want_alignment ? 1 : 0;
  st->slots = std::vector<StreamSlot>((size_t)depth);
  if (const char* t = getenv("DELLYHIP_LOG")) st->log = atoi(t);
  if (const char* t = getenv("DELLYHIP_TEST_FAIL_COLLECT")) st->test_fail_collect = atoi(t);
  if (const char* t = getenv("DELLYHIP_TEST_FAIL_SUBMIT")) st->test_fail_submit = atoi(t);
  st->t_created = now_s();
  {
    DeviceStreams& D = ensure_device_streams(c->device);
    if (!getenv("DELLYHIP_STREAM_ONE_QUEUE")) {
      st->s_up = D.up;
      st->s_down = D.down;
    }
    for (int q = 0; q < 2; ++q) st->s_comp[q] = D.comp[q];
  }
  int slot_index = 0;
  for (auto& S : st->slots) {
    int rc = create_shared_on(c, &c->params, &S.ctx, st->s_comp[slot_index & 1]);
    ++slot_index;
    if (!rc) {
      S.b = new dellyhip_batch();
      hipError_t e = hipEventCreateWithFlags(&S.done, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&S.up_done, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&S.comp_done, hipEventDisableTiming);
      if (e != hipSuccess) rc = fail(DELLYHIP_E_RUNTIME, "hipEventCreate", e);
    }
    if (rc) {
      dellyhip_stream_destroy(st.release());
      return rc;
    }
    // the tuning knobs of the parent (they are read from the environment at dellyhip_create: same values, but a caller
    // may have changed the parent's since)
    S.ctx->sr_sparse = c->sr_sparse; S.ctx->use_sparse = c->use_sparse; S.ctx->use_quad = c->use_quad; S.ctx->quad_mix = c->quad_mix;
    // 12 of the 16 wavefront slots of a CU for a slot's persistent sparse kernel: the next slot's kernel (the other compute
    // stream), the compaction kernels and the memsets find room at once instead of in the kernel's tail -- 32.4 against
    // 29.6 M junctions/s at depth 6 (tools/stream_matrix.sh); DELLYHIP_SPS_WAVES overrides
    // Round 6: with the compact payload (DELLYHIP_COMPACT_ALLELES: 2.9 instead of 9.9 MB down per 10 000 junctions) the room is
    // not needed -- 47.1 against 43.8 M junctions/s with all 16 -- and the cap applies to the full payload only (39.6 / 39.7).
    const bool roomy = depth >= 2 && !getenv("DELLYHIP_SPS_WAVES") && !(c->params.reserved & DELLYHIP_COMPACT_ALLELES);
    S.ctx->sps_waves = roomy ? std::min(c->sps_waves, 12) : c->sps_waves;
    S.ctx->lr_waves = c->lr_waves; S.ctx->lr_teams = c->lr_teams; S.ctx->lr_team_serial = c->lr_team_serial; S.ctx->sparse_cost = c->sparse_cost; S.ctx->msa_tmax = c->msa_tmax;
    S.ctx->msa_waves = c->msa_waves; S.ctx->msa_only = c->msa_only; S.ctx->msa_team = c->msa_team; S.ctx->msa_pair = c->msa_pair; S.ctx->sr_wide = c->sr_wide;
  }
  *out = st.release();
  return 0;
}

void dellyhip_stream_destroy(dellyhip_stream* st) {
  if (!st) return;
  if (st->s_up) (void)hipStreamSynchronize(st->s_up);
  if (st->s_down) (void)hipStreamSynchronize(st->s_down);
  for (auto& S : st->slots) {
    if (S.ctx) {
      (void)hipSetDevice(S.ctx->device);
      (void)hipStreamSynchronize(S.ctx->stream);
    }
    if (S.b) dellyhip_batch_free(S.ctx, S.b);
    if (S.done) (void)hipEventDestroy(S.done);
    if (S.up_done) (void)hipEventDestroy(S.up_done);
    if (S.comp_done) (void)hipEventDestroy(S.comp_done);
    S.in.d.release();
    S.d_rec.release();
    if (S.ctx) dellyhip_destroy(S.ctx);
  }
  delete st;   // (the HIP streams belong to the device: device_streams)
}

// A submit / collect that fails after work has been enqueued must not leave copies or kernels in flight on buffers the next
// submit overwrites, nor a slot that stays "submitted" forever: wait for the device, hand the slot back (the batch is
// dropped; the error is the caller's to report), keep the message of the failure that got us here.
static int slot_bail(StreamSlot& S, int rc) {
  const std::string keep = g_err;
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  S.state = 0;
  S.down_pending = false;
  g_err = keep;
  return rc;
}

int dellyhip_stream_zero_copy(dellyhip_stream* st, int32_t on) {
  if (!st) return fail(DELLYHIP_E_ARG, "null argument");
  st->zero_copy = on != 0;
  return 0;
}

int dellyhip_stream_submit(dellyhip_stream* st, int32_t n, const dellyhip_junction* junc, const char* seq_blob, const uint64_t* seq_off,
                           uint64_t n_seq, uint64_t tag) {
  if (!st) return fail(DELLYHIP_E_ARG, "null argument");
  StreamSlot& S = st->slots[st->n_submit % st->slots.size()];
  if (S.state != 0) return fail(DELLYHIP_E_ARG, "dellyhip_stream_submit: every slot is in flight or held (collect first)");
  dellyhip_ctx* c = S.ctx;
  HIPCHK(hipSetDevice(c->device));
  UploadOpts o;
  o.recycle = S.b;
  o.arena = &S.in;
  o.lazy = true;
  o.up = st->s_up;
  o.up_done = S.up_done;
  o.zero_copy_blob = st->zero_copy;
  dellyhip_batch* b = nullptr;
  const double t0 = now_s();
  int rc = slot_pump(st, nullptr);
  if (rc) return rc;
  rc = batch_upload_impl(c, n, junc, seq_blob, seq_off, n_seq, st->with_msa, st->want_alignment, &b, &o);
  if (rc) return slot_bail(S, rc);   // (the upload may have enqueued part of its copies before it failed)
  // pinned output block: header | records | consensus lengths (msa) | compact blob (at most every slot full)
  S.n = n;
  S.tag = tag;
  S.o_rec = 64;
  S.o_len = S.o_rec + (((size_t)n * sizeof(dellyhip_result) + 63) & ~(size_t)63);
  S.o_blob = S.o_len + (((size_t)n * sizeof(int32_t) + 63) & ~(size_t)63);
  if (st->blob_per_junction <= 0) st->blob_per_junction = st->with_msa ? 1400.0 : 1100.0;
  // capacity: what the estimate asks for with head room; a batch that needs more grows the block at collect time
  S.blob_cap = std::min<uint64_t>((uint64_t)n * b->out_stride, (uint64_t)(st->blob_per_junction * 2.0 * n) + (1u << 16));
  if ((rc = S.out.reserve(S.o_blob + S.blob_cap + 64))) return slot_bail(S, rc);
  if (st->test_fail_submit > 0) { --st->test_fail_submit; return slot_bail(S, fail(DELLYHIP_E_NOMEM, "dellyhip_stream_submit: failure injected by DELLYHIP_TEST_FAIL_SUBMIT")); }
  S.blob_cap = S.out.n - S.o_blob - 64;   // (use what the block has)
  S.blob_cap = std::min<uint64_t>(S.blob_cap, (uint64_t)n * b->out_stride);
  StreamHeader* H = reinterpret_cast<StreamHeader*>(S.out.p);
  H->used = 0;
  H->sps_left = 0;
  b->pin_cons_len = reinterpret_cast<int32_t*>(S.out.p + S.o_len);
  const double t1 = now_s();
  double t2 = t1;
  if (n > 0) {
    if ((rc = dellyhip_batch_run(c, b, nullptr))) return slot_bail(S, rc);
    t2 = now_s();
    if ((rc = slot_compact_and_download(st, S, false))) return slot_bail(S, rc);
  }
  S.state = 1;
  ++st->n_submit;
  if ((rc = slot_pump(st, nullptr))) return rc;
  const double t3 = now_s();
  st->t_stage += t1 - t0;
  st->t_launch += t2 - t1;
  st->t_down += t3 - t2;
  return 0;
}

void dellyhip_stream_release(dellyhip_stream* st) {
  if (st && st->held >= 0) {
    st->slots[st->held].state = 0;
    st->held = -1;
  }
}

void dellyhip_stream_stats(dellyhip_stream* st, double out[6], int32_t reset) {
  if (!st || !out) return;
  out[0] = st->t_stage; out[1] = st->t_launch; out[2] = st->t_down; out[3] = st->t_wait;
  out[4] = (double)st->n_slow; out[5] = (double)st->n_topup;
  if (reset) { st->t_stage = st->t_launch = st->t_down = st->t_wait = 0; st->n_slow = st->n_topup = 0; }
}

int dellyhip_stream_pending(dellyhip_stream* st) {
  return st ? (int)(st->n_submit - st->n_collect) : 0;
}

static int stream_collect_impl(dellyhip_stream* st, const dellyhip_result** results, const char** blob, uint64_t* blob_len, int32_t* n_out,
                               uint64_t* tag) {
  if (!st) return fail(DELLYHIP_E_ARG, "null argument");
  dellyhip_stream_release(st);   // the previous collect()'s output block is released now
  if (st->n_collect == st->n_submit) return fail(DELLYHIP_E_ARG, "dellyhip_stream_collect: nothing submitted");
  const int si = (int)(st->n_collect % st->slots.size());
  StreamSlot& S = st->slots[si];
  dellyhip_ctx* c = S.ctx;
  dellyhip_batch* b = S.b;
  HIPCHK(hipSetDevice(c->device));
  StreamHeader* H = reinterpret_cast<StreamHeader*>(S.out.p);
  if (S.n > 0) {
    const double tw0 = now_s();
    {
      int rcp = slot_pump(st, &S);
      if (rcp) return rcp;
    }
    HIPCHK(hipEventSynchronize(S.done));
    st->t_wait += now_s() - tw0;
    int rc = dellyhip_batch_sync(c, b);
    if (rc) return rc;
    if (st->test_fail_collect > 0) { --st->test_fail_collect; return fail(DELLYHIP_E_NOMEM, "dellyhip_stream_collect: failure injected by DELLYHIP_TEST_FAIL_COLLECT"); }
    if (b->lazy_pending && H->sps_left > 0) {
      ++st->n_slow;
      // rare: junctions beyond the sparse kernel's shapes or level budget -> dense kernels now, then compact again
      if ((rc = finish_lazy(S))) return rc;
      HIPCHK(hipStreamSynchronize(c->stream));
      launch_blob_offsets(c->stream, b->res.p, b->n, b->blob_off.p);
      HIPCHK(hipMemcpyAsync(&H->used, b->blob_off.p + b->n, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (H->used > S.blob_cap) {
        if ((rc = grow_out_block(S, H->used))) return rc;
        H = reinterpret_cast<StreamHeader*>(S.out.p);
      }
      if ((rc = slot_compact_and_download(st, S, true))) return rc;
      HIPCHK(hipEventSynchronize(S.done));
    }
    b->lazy_pending = false;
    if (H->used > S.blob_copied) {   // the estimate was short: fetch the rest
      ++st->n_topup;
      if (H->used > S.blob_cap) {
        const uint64_t used = H->used;
        int rc2 = grow_out_block(S, used);
        if (rc2) return rc2;
        H = reinterpret_cast<StreamHeader*>(S.out.p);
        S.blob_copied = 0;   // (the block moved: records and blob again)
        HIPCHK(hipMemcpyAsync(S.out.p + S.o_rec, S.d_rec.p + 64, (size_t)S.n * sizeof(dellyhip_result), hipMemcpyDeviceToHost, c->stream));
        H->used = used;
      }
      HIPCHK(hipMemcpyAsync(S.out.p + S.o_blob + S.blob_copied, b->blob_compact.p + S.blob_copied, H->used - S.blob_copied,
                            hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      S.blob_copied = H->used;
    }
    const double per = (double)H->used / (double)S.n;
    st->blob_per_junction = (st->n_collect == 0) ? per : 0.75 * st->blob_per_junction + 0.25 * per;
  }
  if (st->log) {
    const uint64_t down = (S.n > 0 ? H->used : 0) + (uint64_t)S.n * sizeof(dellyhip_result);
    st->log_junctions += (uint64_t)S.n;
    st->log_up += S.in.used;
    st->log_down += down;
    time_t tt = time(nullptr);
    struct tm tmv;
    localtime_r(&tt, &tmv);
    char ts[32];
    strftime(ts, sizeof ts, "%Y-%b-%d %H:%M:%S", &tmv);
    fprintf(stderr, "[%s] dellyhip batch %llu: %d junctions, %.0f junctions/s overall, %llu B up, %llu B down (%llu / %llu B in all)\n", ts,
            (unsigned long long)st->n_collect, S.n, (double)st->log_junctions / std::max(now_s() - st->t_created, 1e-9), (unsigned long long)S.in.used,
            (unsigned long long)down, (unsigned long long)st->log_up, (unsigned long long)st->log_down);
  }
  if (results) *results = reinterpret_cast<const dellyhip_result*>(S.out.p + S.o_rec);
  if (blob) *blob = reinterpret_cast<const char*>(S.out.p + S.o_blob);
  if (blob_len) *blob_len = S.n > 0 ? H->used : 0;
  if (n_out) *n_out = S.n;
  if (tag) *tag = S.tag;
  S.state = 2;
  st->held = si;
  ++st->n_collect;
  return 0;
}

int dellyhip_stream_collect(dellyhip_stream* st, const dellyhip_result** results, const char** blob, uint64_t* blob_len, int32_t* n_out,
                            uint64_t* tag) {
  if (!st) return fail(DELLYHIP_E_ARG, "null argument");
  const uint64_t before = st->n_collect;
  const bool had = st->n_collect != st->n_submit;
  const int rc = stream_collect_impl(st, results, blob, blob_len, n_out, tag);
  if (rc && had && st->n_collect == before) {
    // the oldest batch could not be completed (out of memory while growing the output block, a failed dense-kernel pass,
    // a device error): it is dropped -- the slot is free again and the stream goes on with the next batch
    slot_bail(st->slots[before % st->slots.size()], rc);
    ++st->n_collect;
  }
  return rc;
}

// The host-buffer entry points run through a persistent one-slot stream per (mode, want_alignment): staging and device
// buffers survive between calls, so a call costs the copies and the kernels, not hipMalloc / hipFree.
static int run_host_batch(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                          const uint64_t* seq_off, uint64_t n_seq, dellyhip_result* results, char* out_blob,
                          uint64_t cap, uint64_t* used, int with_msa, int want_alignment) {
  if (!c) return fail(DELLYHIP_E_ARG, "null argument");
  if (n < 0 || (n && !results)) return fail(DELLYHIP_E_ARG, "null argument");
  const int key = with_msa * 2 + (want_alignment ? 1 : 0);
  int rc;
  if (!c->host_streams[key] && (rc = dellyhip_stream_create(c, 1, with_msa, want_alignment, &c->host_streams[key]))) return rc;
  dellyhip_stream* st = c->host_streams[key];
  dellyhip_stream_release(st);
  if ((rc = dellyhip_stream_submit(st, n, junc, seq_blob, seq_off, n_seq, 0))) return rc;   // (the slot is free again: slot_bail)
  const dellyhip_result* r = nullptr;
  const char* blob = nullptr;
  uint64_t len = 0;
  rc = dellyhip_stream_collect(st, &r, &blob, &len, nullptr, nullptr);
  if (rc) return rc;
  if (used) *used = len;
  if (len > 0 && (!out_blob || len > cap)) return fail(DELLYHIP_E_ARG, "out_blob too small");
  if (n) memcpy(results, r, (size_t)n * sizeof(dellyhip_result));
  if (len) memcpy(out_blob, blob, len);
  return 0;
}

int dellyhip_align_consensus_batch(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                                   const uint64_t* seq_off, uint64_t n_seq, dellyhip_result* results,
                                   char* out_blob, uint64_t cap, uint64_t* used, int want_alignment) {
  return run_host_batch(c, n, junc, seq_blob, seq_off, n_seq, results, out_blob, cap, used, 0, want_alignment);
}

int dellyhip_refine_batch(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                          const uint64_t* seq_off, uint64_t n_seq, dellyhip_result* results, char* out_blob,
                          uint64_t cap, uint64_t* used, int want_alignment) {
  return run_host_batch(c, n, junc, seq_blob, seq_off, n_seq, results, out_blob, cap, used, 1, want_alignment);
}

int dellyhip_refine_batch_lr(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                             const uint64_t* seq_off, uint64_t n_seq, dellyhip_result* results, char* out_blob,
                             uint64_t cap, uint64_t* used, int want_alignment) {
  return run_host_batch(c, n, junc, seq_blob, seq_off, n_seq, results, out_blob, cap, used, 2, want_alignment);
}

int dellyhip_msa_edlib(dellyhip_ctx* c, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off, char* cs,
                       int32_t cs_cap, int32_t* cs_len, int32_t* rows) {
  if (!c || !cs_len || !rows || n_reads < 0 || (n_reads && (!seq_blob || !seq_off))) return fail(DELLYHIP_E_ARG, "bad argument");
  if (n_reads > dh::LM_NR) return fail(DELLYHIP_E_LIMIT, "msaEdlib: more reads than the kernel holds");
  HIPCHK(hipSetDevice(c->device));
  *cs_len = 0;
  *rows = 0;
  if (n_reads == 0) return 0;
  int rc;
  const uint64_t blob_bytes = seq_off[n_reads];
  int maxlen = 1;
  for (int k = 0; k < n_reads; ++k) maxlen = std::max<int>(maxlen, (int)(seq_off[k + 1] - seq_off[k]));
  if (maxlen > dh::LR_NMAX) return fail(DELLYHIP_E_LIMIT, "msaEdlib: read longer than the kernel limit");
  dellyhip_junction J{};
  J.svt = 2;
  J.n_seq = n_reads;
  J.seq_first = 0;
  const int32_t pf[2] = {0, n_reads * (n_reads - 1) / 2};
  DevBuf<dellyhip_junction> dj;
  DevBuf<uint8_t> dblob, dout, dws;
  DevBuf<uint64_t> doff;
  DevBuf<int32_t> dpf, dedit, dlen;
  DevBuf<dellyhip_result> dres;
  dh::LrMsaArgs M{};
  lm_layout(M, maxlen);
  int nlong = 0;
  for (int k = 0; k < n_reads; ++k) nlong += (seq_off[k + 1] - seq_off[k] > (uint64_t)dh::MYERS_ROWS) ? 1 : 0;
  const uint64_t hhalf = (nlong >= 2) ? (((uint64_t)maxlen + 16 + 255) & ~255ull) : 0;
  DevBuf<int8_t> dhb;
  if (hhalf && (rc = dhb.alloc((size_t)2 * hhalf * std::max(pf[1], 1)))) return rc;
  if ((rc = dj.alloc(1)) || (rc = dblob.alloc(blob_bytes + 64)) || (rc = doff.alloc(n_reads + 1)) ||
      (rc = dpf.alloc(2)) || (rc = dedit.alloc(dh::LM_NR * dh::LM_NR)) || (rc = dlen.alloc(1)) || (rc = dres.alloc(1)) ||
      (rc = dout.alloc(M.acap)) || (rc = dws.alloc(M.ws_stride)))
    return rc;
  HIPCHK(hipMemcpy(dj.p, &J, sizeof J, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dblob.p, seq_blob, blob_bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(doff.p, seq_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dpf.p, pf, sizeof pf, hipMemcpyHostToDevice));
  HIPCHK(hipMemsetAsync(dres.p, 0, sizeof(dellyhip_result), c->stream));   // (on the launch stream: see batch_upload_impl)
  HIPCHK(hipMemsetAsync(dedit.p, 0, dh::LM_NR * dh::LM_NR * sizeof(int32_t), c->stream));
  if (pf[1] > 0) {
    dh::PairArgs pa{dj.p, dblob.p, doff.p, dpf.p, 1, pf[1], dh::LM_NR, dedit.p, dhb.p, hhalf, 0, 0, 0};
    hipLaunchKernelGGL(dh::myers_pairs_kernel, dim3(pf[1]), dim3(dh::WAVE), 0, c->stream, pa);
    HIPCHK(hipGetLastError());
  }
  M.junc = dj.p; M.seq_blob = dblob.p; M.seq_off = doff.p; M.p = c->params; M.res = dres.p;
  M.out_blob = dout.p; M.out_stride = M.acap; M.out_cons_cap = M.acap; M.cons_len = dlen.p; M.edit = dedit.p;
  M.n_work = 1; M.ws = dws.p;
  hipLaunchKernelGGL(dh::lrmsa_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, M);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  dellyhip_result R{};
  int32_t L = 0;
  HIPCHK(hipMemcpy(&R, dres.p, sizeof R, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&L, dlen.p, sizeof L, hipMemcpyDeviceToHost));
  if (R.status) return fail(R.status, "msaEdlib: kernel limit");
  *rows = R.sr_support;
  *cs_len = L;
  if (L > cs_cap) return fail(DELLYHIP_E_ARG, "consensus buffer too small");
  if (L > 0) HIPCHK(hipMemcpy(cs, dout.p, L, hipMemcpyDeviceToHost));
  return 0;
}

int dellyhip_msa_wfa(dellyhip_ctx* c, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off, const char* prefix,
                     int32_t prefix_len, const char* suffix, int32_t suffix_len, char* cs, int32_t cs_cap, int32_t* cs_len,
                     int32_t* rows) {
  if (!c || !cs_len || !rows || n_reads < 0 || (n_reads && (!seq_blob || !seq_off)) || prefix_len < 0 || suffix_len < 0 ||
      (prefix_len && !prefix) || (suffix_len && !suffix))
    return fail(DELLYHIP_E_ARG, "bad argument");
  if (n_reads > dh::LM_NR || prefix_len > dh::WFA_PCAP || suffix_len > dh::WFA_PCAP)
    return fail(DELLYHIP_E_LIMIT, "msaWfa: more reads / longer anchors than the kernel holds");
  HIPCHK(hipSetDevice(c->device));
  *cs_len = 0;
  *rows = 0;
  if (n_reads == 0) return 0;
  int rc;
  const uint64_t blob_bytes = seq_off[n_reads];
  int maxlen = 1;
  for (int k = 0; k < n_reads; ++k) maxlen = std::max<int>(maxlen, (int)(seq_off[k + 1] - seq_off[k]));
  if (maxlen > dh::LR_NMAX) return fail(DELLYHIP_E_LIMIT, "msaWfa: read longer than the kernel limit");
  dellyhip_junction J{};
  J.svt = 4;
  J.n_seq = n_reads;
  J.seq_first = 0;
  DevBuf<dellyhip_junction> dj;
  DevBuf<uint8_t> dblob, dout, dws, dpre, dsuf;
  DevBuf<uint64_t> doff;
  DevBuf<int32_t> dlen;
  DevBuf<dellyhip_result> dres;
  dh::LrWfaArgs W{};
  wfa_layout(W, maxlen);
  if ((rc = dj.alloc(1)) || (rc = dblob.alloc(blob_bytes + 64)) || (rc = doff.alloc(n_reads + 1)) ||
      (rc = dlen.alloc(1)) || (rc = dres.alloc(1)) || (rc = dout.alloc(W.acap)) || (rc = dws.alloc(W.ws_stride)) ||
      (rc = dpre.alloc(std::max(prefix_len, 1))) || (rc = dsuf.alloc(std::max(suffix_len, 1))))
    return rc;
  HIPCHK(hipMemcpy(dj.p, &J, sizeof J, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dblob.p, seq_blob, blob_bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(doff.p, seq_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  if (prefix_len) HIPCHK(hipMemcpy(dpre.p, prefix, prefix_len, hipMemcpyHostToDevice));
  if (suffix_len) HIPCHK(hipMemcpy(dsuf.p, suffix, suffix_len, hipMemcpyHostToDevice));
  HIPCHK(hipMemsetAsync(dres.p, 0, sizeof(dellyhip_result), c->stream));   // (on the launch stream: see batch_upload_impl)
  HIPCHK(hipMemsetAsync(dws.p, 0, W.ws_stride, c->stream));
  W.junc = dj.p; W.seq_blob = dblob.p; W.seq_off = doff.p; W.p = c->params; W.res = dres.p;
  W.out_blob = dout.p; W.out_stride = W.acap; W.out_cons_cap = W.acap; W.cons_len = dlen.p;
  W.work_list = nullptr; W.n_work = 1; W.use_anchors = 0;
  W.prefix = dpre.p; W.suffix = dsuf.p; W.prefix_len = prefix_len; W.suffix_len = suffix_len;
  W.ws = dws.p;
  hipLaunchKernelGGL(dh::lrwfa_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, W);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  dellyhip_result R{};
  int32_t L = 0;
  HIPCHK(hipMemcpy(&R, dres.p, sizeof R, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&L, dlen.p, sizeof L, hipMemcpyDeviceToHost));
  if (R.status) return fail(R.status, "msaWfa: kernel limit");
  *rows = R.sr_support;
  *cs_len = L;
  if (L > cs_cap) return fail(DELLYHIP_E_ARG, "consensus buffer too small");
  if (L > 0) HIPCHK(hipMemcpy(cs, dout.p, L, hipMemcpyDeviceToHost));
  return 0;
}

// one direct (consensus, window) pair through the short-read kernels: svt 2 = longNeedle, svt 4 = splitAlign
static int direct_pair(dellyhip_ctx* c, int svt, const char* s1, int32_t m, const char* s2, int32_t n, char* align_rows,
                       int32_t aln_cap, int32_t* aln_len, int32_t* found, const char* what) {
  if (!c || !s1 || !s2 || !aln_len || !found || m < 0 || n < 0) return fail(DELLYHIP_E_ARG, "bad argument");
  // beyond the short-read kernels' shapes: the long-read kernels in their direct mode (round 6; the reference has no limit,
  // src/needle.h:45-47, src/split.h:480-482)
  const bool big = m > dh::MMAX || n > dh::NMAX;
  if (big && (m > dh::LR_MMAX || n > dh::LR_NMAX)) return fail(DELLYHIP_E_LIMIT, what);
  HIPCHK(hipSetDevice(c->device));
  dellyhip_batch* b = new dellyhip_batch();
  b->n = 1;
  b->want_alignment = 1;
  b->out_stride = (dh::OUT_CONS_CAP + dh::OUT_ALLELE_CAP + dh::OUT_ALN_CAP + 15) & ~15ull;
  if (big) {   // output slot of the long-read shapes (as batch_upload_impl sizes it)
    b->out_cons_cap = std::max<int>(dh::OUT_CONS_CAP, (m + 16) & ~15);
    b->out_allele_cap = std::max<int>(dh::OUT_ALLELE_CAP, (m + n + 8 + 15) & ~15);
    b->out_aln_cap = std::max<int>(dh::OUT_ALN_CAP, 2 * ((m + n + 8 + 15) & ~15));
    b->out_stride = ((uint64_t)b->out_cons_cap + b->out_allele_cap + b->out_aln_cap + 15) & ~15ull;
    b->h_win_len.assign(1, n);
  }
  dellyhip_junction J{};
  J.svt = svt;
  J.n_seq = 1;
  b->h_junc.assign(1, J);
  b->h_cons_len.assign(1, m);
  int rc = 0;
  auto bail = [&](int r) { dellyhip_batch_free(c, b); return r; };
  uint64_t zero = 0;
  if ((rc = b->junc.alloc(1)) || (rc = b->seq_blob.alloc(std::max(m, 1))) || (rc = b->cons_off.alloc(1)) ||
      (rc = b->cons_len.alloc(1)) || (rc = b->res.alloc(1)) || (rc = b->out_blob.alloc(b->out_stride)) ||
      (rc = b->ref_blob.alloc(std::max(n, 1))) || (rc = b->ref_off.alloc(1)) || (rc = b->ref_len.alloc(1)))
    return bail(rc);
  (void)hipMemcpy(b->junc.p, &J, sizeof J, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->seq_blob.p, s1, m, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->cons_off.p, &zero, 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->cons_len.p, &m, 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->ref_blob.p, s2, n, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->ref_off.p, &zero, 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(b->ref_len.p, &n, 4, hipMemcpyHostToDevice);
  (void)hipMemsetAsync(b->res.p, 0, sizeof(dellyhip_result), c->stream);
  (void)hipStreamSynchronize(c->stream);
  if (big && svt == 4 && (rc = setup_lri_workspace(c, b, m, n, 1))) return bail(rc);
  if (big && svt != 4 && (rc = setup_lr_workspace(c, b, m, n, 1))) return bail(rc);
  if ((rc = build_bins(b, c->params))) return bail(rc);
  rc = dellyhip_batch_run(c, b, nullptr);
  dellyhip_result R;
  std::vector<char> blob(b->out_stride);
  uint64_t used = 0;
  if (!rc) rc = dellyhip_batch_fetch(c, b, &R, blob.data(), blob.size(), &used);
  if (!rc) {
    if (R.status) rc = fail(R.status, what);
    else {
      *found = R.ok;
      *aln_len = R.ok ? R.aln_len : 0;
      if (R.ok) {
        if (R.aln_len > aln_cap || !align_rows) rc = fail(DELLYHIP_E_ARG, "align_rows too small");
        else {
          memcpy(align_rows, blob.data() + R.aln_off, R.aln_len);
          memcpy(align_rows + aln_cap, blob.data() + R.aln_off + R.aln_len, R.aln_len);
        }
      }
    }
  }
  dellyhip_batch_free(c, b);
  return rc;
}

int dellyhip_long_needle(dellyhip_ctx* c, const char* s1, int32_t m, const char* s2, int32_t n, char* align_rows,
                         int32_t aln_cap, int32_t* aln_len, int32_t* found) {
  return direct_pair(c, 2, s1, m, s2, n, align_rows, aln_cap, aln_len, found, "longNeedle operand exceeds the short-read kernel limits");
}

int dellyhip_split_align(dellyhip_ctx* c, const char* cons, int32_t m, const char* ref, int32_t n, char* align_rows,
                         int32_t aln_cap, int32_t* aln_len, int32_t* found) {
  if (m < 1 || n < 3) return fail(DELLYHIP_E_LIMIT, "splitAlign: |cons| >= 1 and |svRefStr| >= 3 (src/split.h:513-517 indexes distRev[n-2])");
  return direct_pair(c, 4, cons, m, ref, n, align_rows, aln_cap, aln_len, found, "splitAlign operand exceeds the short-read kernel limits");
}

int dellyhip_edlib_align(dellyhip_ctx* c, const char* query, int32_t qn, const char* target, int32_t tn, int32_t mode,
                         int32_t task, int32_t out[4], unsigned char* ops, int32_t ops_cap, int32_t* ops_len) {
  if (!c || !out || !ops_len || qn < 0 || tn < 0 || (qn && !query) || (tn && !target) || mode < 0 || mode > 2 ||
      task < 0 || task > 2)
    return fail(DELLYHIP_E_ARG, "bad argument");
  *ops_len = 0;
  if (qn == 0 || tn == 0) {  // edlib.cpp:160-178: no locations beyond the end, no alignment
    out[0] = (mode == 0) ? std::max(qn, tn) : qn;
    out[1] = 1;
    out[2] = (mode == 0) ? tn - 1 : -1;
    out[3] = -2;
    return 0;
  }
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if (mode == 0 && task == 0 && std::min(qn, tn) <= dh::MYERS_ROWS) {
    // NW distance: Myers bit-vector kernel, pattern = the shorter string (the distance is symmetric)
    DevBuf<uint8_t> dq, dt;
    DevBuf<int32_t> dout;
    if ((rc = dq.alloc(qn)) || (rc = dt.alloc(tn)) || (rc = dout.alloc(2))) return rc;
    HIPCHK(hipMemcpy(dq.p, query, qn, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dt.p, target, tn, hipMemcpyHostToDevice));
    if (qn <= tn) hipLaunchKernelGGL(dh::myers_single_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, dq.p, qn, dt.p, tn, dout.p);
    else hipLaunchKernelGGL(dh::myers_single_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, dt.p, tn, dq.p, qn, dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    int32_t d = 0;
    HIPCHK(hipMemcpy(&d, dout.p, sizeof d, hipMemcpyDeviceToHost));
    out[0] = d;
    out[1] = 1;
    out[2] = tn - 1;
    out[3] = -2;
    return 0;
  }
  if (tn > dh::MMAX || qn > dh::NMAX) {
    // beyond the insertion kernel's shapes: the strip machinery of dellyhip_edlib_align_full (any shape it takes; round 6 --
    // the reference's edlibAlign has no limit).  out[] as above: distance, locations, first end, first start (-2: none)
    std::vector<int32_t> ends((size_t)tn + 2), starts((size_t)tn + 2);
    int32_t ed = -1, nloc = 0;
    rc = dellyhip_edlib_align_full(c, query, qn, target, tn, -1, mode, task, 0, &ed, &nloc, ends.data(), starts.data(), (int32_t)ends.size(), ops, ops_cap, ops_len);
    if (rc) return rc;
    out[0] = ed;
    out[1] = nloc;
    out[2] = nloc > 0 ? ends[0] : -1;
    out[3] = (task >= 1 && nloc > 0) ? starts[0] : -2;
    return 0;
  }
  if ((rc = ensure_scratch(c))) return rc;
  DevBuf<uint8_t> dq, dt, dops;
  DevBuf<int32_t> dout;
  if ((rc = dq.alloc(qn)) || (rc = dt.alloc(tn)) || (rc = dops.alloc(qn + tn + 8)) || (rc = dout.alloc(8))) return rc;
  HIPCHK(hipMemcpy(dq.p, query, qn, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, target, tn, hipMemcpyHostToDevice));
  dh::EdArgs a{dq.p, dt.p, qn, tn, mode, task, dout.p, dops.p, c->scratch.p};
  hipLaunchKernelGGL(dh::edlib_single_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  int32_t h[5];
  HIPCHK(hipMemcpy(h, dout.p, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
  *ops_len = h[4];
  if (h[4] > 0) {
    if (!ops || h[4] > ops_cap) return fail(DELLYHIP_E_ARG, "ops buffer too small");
    HIPCHK(hipMemcpy(ops, dops.p, h[4], hipMemcpyDeviceToHost));
  }
  return 0;
}

// edlibAlign with everything its result struct holds, any shape the strip machinery takes (edlib_kernel.hpp)
int dellyhip_edlib_align_full(dellyhip_ctx* c, const char* query, int32_t qn, const char* target, int32_t tn, int32_t k, int32_t mode,
                              int32_t task, int32_t equalities, int32_t* edit_distance, int32_t* num_locations, int32_t* end_locs,
                              int32_t* start_locs, int32_t loc_cap, unsigned char* ops, int32_t ops_cap, int32_t* ops_len) {
  if (!c || !edit_distance || !num_locations || !ops_len || qn < 0 || tn < 0 || (qn && !query) || (tn && !target) || mode < 0 ||
      mode > 2 || task < 0 || task > 2 || equalities < 0 || equalities > 1 || loc_cap < 0 || (loc_cap && !end_locs) ||
      (task >= 1 && loc_cap && !start_locs))
    return fail(DELLYHIP_E_ARG, "bad argument");
  *ops_len = 0;
  *num_locations = 0;
  *edit_distance = -1;
  if (qn == 0 || tn == 0) {  // edlib.cpp:160-178: one location, no start locations, no alignment (k is not looked at)
    if (loc_cap < 1) return fail(DELLYHIP_E_ARG, "loc_cap");
    *edit_distance = (mode == 0) ? std::max(qn, tn) : qn;
    *num_locations = 1;
    end_locs[0] = (mode == 0) ? tn - 1 : -1;
    return 0;
  }
  if (tn > dh::LM_RMASK - 1 || qn > dh::LR_NMAX) return fail(DELLYHIP_E_LIMIT, "edlibAlign: target beyond 32766 or query beyond 32000 letters");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  dh::EdFullArgs a{};
  a.qn = qn; a.tn = tn; a.mode = mode; a.task = task; a.eq = equalities;
  a.bnd_stride = qn + 128;
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = o; o += (bytes + 255) & ~255ull; return at; };
  take(4ull * (uint64_t)a.bnd_stride * 4);
  a.off_tmp = take((uint64_t)qn + tn + 64);
  a.off_lastcol = take(((uint64_t)tn + 2) * 4);
  a.strip_words = (task == 2) ? dh::lm_dirs_words(tn, qn) : 64;
  a.off_dirs = take(a.strip_words * 4);
  const int32_t lcap = (mode == 0) ? 1 : tn + 1;
  DevBuf<uint8_t> dq, dt, dops, ws;
  DevBuf<int32_t> dout, dend, dstart;
  if ((rc = dq.alloc(qn + 16)) || (rc = dt.alloc(tn + 16)) || (rc = dops.alloc((uint64_t)qn + tn + 64)) || (rc = ws.alloc(o)) || (rc = dout.alloc(8)) ||
      (rc = dend.alloc(lcap)) || (rc = dstart.alloc(lcap)))
    return rc;
  HIPCHK(hipMemcpyAsync(dq.p, query, qn, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dt.p, target, tn, hipMemcpyHostToDevice, c->stream));
  a.q = dq.p; a.t = dt.p; a.ws = ws.p; a.out = dout.p; a.end_locs = dend.p; a.start_locs = dstart.p; a.loc_cap = lcap;
  a.ops = dops.p; a.ops_cap = qn + tn + 32;
  hipLaunchKernelGGL(dh::edlib_full_kernel, dim3(1), dim3(dh::WAVE), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  int32_t h[4];
  HIPCHK(hipMemcpyAsync(h, dout.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (h[3]) return fail(DELLYHIP_E_LIMIT, "edlibAlign: the alignment exceeds the kernel's workspace");
  if (k >= 0 && h[0] > k) return 0;   // editDistance = -1, no locations, no alignment (src/edlib.h: "-1 if ... larger than k")
  *edit_distance = h[0];
  *num_locations = h[1];
  if (h[1] > loc_cap) return fail(DELLYHIP_E_ARG, "edlibAlign: loc_cap too small (needed count in *num_locations)");
  if (h[1] > 0) {
    HIPCHK(hipMemcpy(end_locs, dend.p, (size_t)h[1] * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (task >= 1) HIPCHK(hipMemcpy(start_locs, dstart.p, (size_t)h[1] * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  *ops_len = h[2];
  if (h[2] > 0) {
    if (!ops || h[2] > ops_cap) return fail(DELLYHIP_E_ARG, "ops buffer too small");
    HIPCHK(hipMemcpy(ops, dops.p, h[2], hipMemcpyDeviceToHost));
  }
  return 0;
}

// ---- split-read genotyping classifier (src/coverage.h:412-434), SURVEY.md 8f N1 ----------------------
static_assert(sizeof(dellyhip_align_job) == 48 && sizeof(dellyhip_align_result) == 20, "C-ABI record layout");
struct dellyhip_jobs {
  uint64_t n = 0;
  DevBuf<dellyhip_align_job> jobs;
  DevBuf<uint8_t> blob;
  DevBuf<dellyhip_align_result> res;
  DevBuf<int32_t> wide;      // [0] = count, [1..] = job indices with a probe > 64 bytes
  DevBuf<int32_t> big;       // [0] = count, [1..] = job indices with a probe > 256 bytes
  hipStream_t last_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms_sum = 0;
  int launches = 0;
  ~dellyhip_jobs() {
    for (auto& e : pending) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  }
};

int dellyhip_jobs_upload(dellyhip_ctx* c, uint64_t n_jobs, const dellyhip_align_job* jobs, const char* blob,
                         uint64_t blob_len, dellyhip_jobs** out) {
  if (!c || !out || (n_jobs && !jobs) || (blob_len && !blob) || n_jobs >= (1ull << 31))
    return fail(DELLYHIP_E_ARG, "bad argument");
  for (uint64_t i = 0; i < n_jobs; ++i) {
    const dellyhip_align_job& j = jobs[i];
    if (j.cons_off + j.cons_len > blob_len || j.ref_off + j.ref_len > blob_len || j.seq_off + j.seq_len > blob_len ||
        j.seq_len > 0x7fffffffu)
      return fail(DELLYHIP_E_ARG, "align job points outside the blob");
  }
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<dellyhip_jobs> b(new dellyhip_jobs);
  b->n = n_jobs;
  int rc;
  if ((rc = b->jobs.alloc(std::max<uint64_t>(n_jobs, 1))) || (rc = b->blob.alloc(blob_len + 2 * dh::CLS_PAD)) ||
      (rc = b->res.alloc(std::max<uint64_t>(n_jobs, 1))) || (rc = b->wide.alloc(n_jobs + 1)) || (rc = b->big.alloc(n_jobs + 1)))
    return rc;
  if (n_jobs) HIPCHK(hipMemcpyAsync(b->jobs.p, jobs, n_jobs * sizeof(dellyhip_align_job), hipMemcpyHostToDevice, c->stream));
  if (blob_len) HIPCHK(hipMemcpyAsync(b->blob.p, blob, blob_len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(b->blob.p + blob_len, 0, 2 * dh::CLS_PAD, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = b.release();
  return 0;
}

int dellyhip_jobs_run(dellyhip_ctx* c, dellyhip_jobs* b, void* stream_) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = stream_ ? (hipStream_t)stream_ : c->stream;
  b->last_stream = st;
  if (b->n == 0) return 0;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipMemsetAsync(b->wide.p, 0, sizeof(int32_t), st));
  HIPCHK(hipMemsetAsync(b->big.p, 0, sizeof(int32_t), st));
  dh::ClsArgs a{b->jobs.p, b->blob.p, b->res.p, b->n, c->params.flank_quality, b->wide.p + 1, b->wide.p, b->big.p + 1, b->big.p};
  const uint64_t groups = (b->n + dh::WAVE - 1) / dh::WAVE;
  const int grid = (int)std::min<uint64_t>(groups, (uint64_t)std::max(1, c->n_cu) * 64);
  HIPCHK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(dh::classify_kernel<1>, dim3(grid), dim3(dh::WAVE), 0, st, a);
  HIPCHK(hipEventRecord(e1, st));
  // probes of 65 .. 256 bytes (the list is usually empty: the launch then costs a few microseconds)
  hipLaunchKernelGGL(dh::classify_kernel<dh::CLS_MAXW>, dim3(std::max(1, c->n_cu)), dim3(dh::WAVE), 0, st, a);
  // probes beyond 256 bytes (one job per wavefront; the list is practically always empty)
  hipLaunchKernelGGL(dh::classify_big_kernel, dim3(std::max(1, c->n_cu)), dim3(dh::WAVE), 0, st, a);
  HIPCHK(hipGetLastError());
  b->pending.emplace_back(e0, e1);
  return 0;
}

int dellyhip_jobs_sync(dellyhip_ctx* c, dellyhip_jobs* b) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(b->last_stream ? b->last_stream : c->stream));
  return 0;
}

int dellyhip_jobs_kernel_ms(dellyhip_ctx* c, dellyhip_jobs* b, double* ms, int32_t* launches) {
  if (!c || !b || !ms || !launches) return fail(DELLYHIP_E_ARG, "bad argument");
  int rc = dellyhip_jobs_sync(c, b);
  if (rc) return rc;
  double sum = 0;
  for (auto& e : b->pending) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, e.first, e.second));
    sum += t;
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  *launches = (int32_t)b->pending.size();
  *ms = b->pending.empty() ? 0.0 : sum / (double)b->pending.size();
  b->pending.clear();
  return 0;
}

int dellyhip_jobs_fetch(dellyhip_ctx* c, dellyhip_jobs* b, dellyhip_align_result* results) {
  if (!c || !b || (b->n && !results)) return fail(DELLYHIP_E_ARG, "bad argument");
  int rc = dellyhip_jobs_sync(c, b);
  if (rc) return rc;
  if (b->n) HIPCHK(hipMemcpy(results, b->res.p, b->n * sizeof(dellyhip_align_result), hipMemcpyDeviceToHost));
  return 0;
}

void dellyhip_jobs_free(dellyhip_ctx* c, dellyhip_jobs* b) {
  if (!b) return;
  if (c) hipSetDevice(c->device);
  delete b;
}

int dellyhip_classify_reads(dellyhip_ctx* c, uint64_t n_jobs, const dellyhip_align_job* jobs, const char* blob,
                            uint64_t blob_len, dellyhip_align_result* results) {
  dellyhip_jobs* b = nullptr;
  int rc = dellyhip_jobs_upload(c, n_jobs, jobs, blob, blob_len, &b);
  if (rc) return rc;
  rc = dellyhip_jobs_run(c, b, nullptr);
  if (!rc) rc = dellyhip_jobs_fetch(c, b, results);
  dellyhip_jobs_free(c, b);
  return rc;
}


// ---- long-read genotyping: batched _editDistanceNW (src/genotype.h:21-30), SURVEY.md 8f N2 -----------------
static_assert(sizeof(dellyhip_nw_job) == 24, "C-ABI record layout");
struct dellyhip_nwjobs {
  uint64_t n = 0;
  DevBuf<dellyhip_nw_job> jobs;
  DevBuf<uint8_t> blob;
  DevBuf<int32_t> dist;
  DevBuf<uint32_t> next;
  DevBuf<int8_t> hbuf;          // strip passes of pairs with both strings beyond MYERS_ROWS
  uint64_t hbuf_half = 0;
  int grid = 1;
  int pairwise = 0;             // adjacent jobs side by side in one wavefront (most pairs gain from it)
  hipStream_t last_stream = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  ~dellyhip_nwjobs() {
    for (auto& e : pending) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  }
};

int dellyhip_nwjobs_upload(dellyhip_ctx* c, uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob,
                           uint64_t blob_len, dellyhip_nwjobs** out) {
  if (!c || !out || (n_jobs && !jobs) || (blob_len && !blob)) return fail(DELLYHIP_E_ARG, "bad argument");
  for (uint64_t i = 0; i < n_jobs; ++i)
    if (jobs[i].query_off + jobs[i].query_len > blob_len || jobs[i].target_off + jobs[i].target_len > blob_len ||
        jobs[i].query_len > 0x7fffffffu || jobs[i].target_len > 0x7fffffffu)
      return fail(DELLYHIP_E_ARG, "nw job points outside the blob");
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<dellyhip_nwjobs> b(new dellyhip_nwjobs);
  b->n = n_jobs;
  int rc;
  if ((rc = b->jobs.alloc(std::max<uint64_t>(n_jobs, 1))) || (rc = b->blob.alloc(blob_len + 64)) ||
      (rc = b->dist.alloc(std::max<uint64_t>(n_jobs, 1))) || (rc = b->next.alloc(1)))
    return rc;
  if (n_jobs >= (1ull << 31) - 65536) return fail(DELLYHIP_E_ARG, "too many nw jobs");
  b->grid = (int)std::min<uint64_t>(std::max<uint64_t>(n_jobs, 1), (uint64_t)std::max(1, c->n_cu) * 28);   // 7 wavefronts per SIMD (71 VGPRs)
  uint64_t longest = 0;
  for (uint64_t i = 0; i < n_jobs; ++i)
    if (jobs[i].query_len > (uint32_t)dh::MYERS_ROWS && jobs[i].target_len > (uint32_t)dh::MYERS_ROWS)
      longest = std::max<uint64_t>(longest, std::max(jobs[i].query_len, jobs[i].target_len));
  {
    uint64_t pays = 0;
    for (uint64_t i = 0; i + 1 < n_jobs; i += 2)
      pays += dh::myers_x2_pays((int)std::min(jobs[i].query_len, jobs[i].target_len), (int)std::min(jobs[i + 1].query_len, jobs[i + 1].target_len)) ? 1 : 0;
    b->pairwise = (n_jobs >= 2 && pays * 10 >= (n_jobs / 2) * 6) ? 1 : 0;
  }
  if (longest) {
    b->hbuf_half = (longest + 16 + 255) & ~255ull;
    if ((rc = b->hbuf.alloc((size_t)2 * b->hbuf_half * b->grid))) return rc;
  }
  if (n_jobs) HIPCHK(hipMemcpyAsync(b->jobs.p, jobs, n_jobs * sizeof(dellyhip_nw_job), hipMemcpyHostToDevice, c->stream));
  if (blob_len) HIPCHK(hipMemcpyAsync(b->blob.p, blob, blob_len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = b.release();
  return 0;
}

int dellyhip_nwjobs_run(dellyhip_ctx* c, dellyhip_nwjobs* b, void* stream_) {
  if (!c || !b) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = stream_ ? (hipStream_t)stream_ : c->stream;
  b->last_stream = st;
  if (b->n == 0) return 0;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  dh::NwArgs a{b->jobs.p, b->blob.p, b->dist.p, b->n, b->next.p, b->hbuf.p, b->hbuf_half, b->pairwise};
  const int grid = b->grid;
  HIPCHK(hipMemsetAsync(b->next.p, 0, sizeof(uint32_t), st));
  HIPCHK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(dh::nw_jobs_kernel, dim3(grid), dim3(dh::WAVE), 0, st, a);
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipGetLastError());
  b->pending.emplace_back(e0, e1);
  return 0;
}

int dellyhip_nwjobs_kernel_ms(dellyhip_ctx* c, dellyhip_nwjobs* b, double* ms, int32_t* launches) {
  if (!c || !b || !ms || !launches) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(b->last_stream ? b->last_stream : c->stream));
  double sum = 0;
  for (auto& e : b->pending) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, e.first, e.second));
    sum += t;
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  *launches = (int32_t)b->pending.size();
  *ms = b->pending.empty() ? 0.0 : sum / (double)b->pending.size();
  b->pending.clear();
  return 0;
}

int dellyhip_nwjobs_fetch(dellyhip_ctx* c, dellyhip_nwjobs* b, int32_t* distances) {
  if (!c || !b || (b->n && !distances)) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(b->last_stream ? b->last_stream : c->stream));
  if (b->n) HIPCHK(hipMemcpy(distances, b->dist.p, b->n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}

void dellyhip_nwjobs_free(dellyhip_ctx* c, dellyhip_nwjobs* b) {
  if (!b) return;
  if (c) hipSetDevice(c->device);
  delete b;
}

int dellyhip_edit_distance_nw_batch(dellyhip_ctx* c, uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob,
                                    uint64_t blob_len, int32_t* distances) {
  dellyhip_nwjobs* b = nullptr;
  int rc = dellyhip_nwjobs_upload(c, n_jobs, jobs, blob, blob_len, &b);
  if (rc) return rc;
  rc = dellyhip_nwjobs_run(c, b, nullptr);
  if (!rc) rc = dellyhip_nwjobs_fetch(c, b, distances);
  dellyhip_nwjobs_free(c, b);
  return rc;
}


// ---- probe generation (src/coverage.h:164-263), SURVEY.md 8f N3 ------------------------------------------
static_assert(sizeof(dellyhip_probes) == 96, "C-ABI record layout");

int dellyhip_batch_probes(dellyhip_ctx* c, dellyhip_batch* b, dellyhip_probes* probes, char* out_blob, uint64_t cap,
                          uint64_t* used) {
  if (!c || !b || (b->n && !probes) || !used) return fail(DELLYHIP_E_ARG, "bad argument");
  if (b->with_msa) return fail(DELLYHIP_E_ARG, "probes need a batch with the consensus given (with_msa = 0)");
  if (!b->ever_run) return fail(DELLYHIP_E_ARG, "run the batch first");
  int rc = dellyhip_batch_sync(c, b);
  if (rc) return rc;
  *used = 0;
  if (b->n == 0) return 0;
  HIPCHK(hipSetDevice(c->device));
  if ((rc = ensure_chr_table(c))) return rc;
  DevBuf<dellyhip_probes> d_rec;
  DevBuf<uint8_t> d_blob;
  const size_t slot = 4 * (size_t)dh::PROBE_CAP;
  if ((rc = d_rec.alloc(b->n)) || (rc = d_blob.alloc(slot * b->n))) return rc;
  dh::ProbeArgs A{};
  A.a.junc = b->junc.p;
  A.a.cons_base = b->seq_blob.p;
  A.a.cons_off = b->cons_off.p;
  A.a.cons_len = b->cons_len.p;
  A.a.chr_seq = c->d_chr_ptr.p;
  A.a.chr_len = c->d_chr_len.p;
  A.a.n_chr = (int)c->chr_dev.size();
  A.a.p = c->params;
  A.a.res = b->res.p;
  A.out = d_rec.p;
  A.blob = d_blob.p;
  A.n = b->n;
  hipLaunchKernelGGL(dh::probes_kernel, dim3(std::min(b->n, std::max(1, c->n_cu) * 16)), dim3(dh::WAVE), 0, c->stream, A);
  HIPCHK(hipGetLastError());
  std::vector<uint8_t> h_blob(slot * b->n);
  HIPCHK(hipMemcpyAsync(probes, d_rec.p, sizeof(dellyhip_probes) * b->n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(h_blob.data(), d_blob.p, h_blob.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t pos = 0;
  for (int j = 0; j < b->n; ++j) {   // pack the fixed-stride slots
    dellyhip_probes& P = probes[j];
    if (!P.ok) continue;
    for (int bp = 0; bp < 2; ++bp) {
      for (int which = 0; which < 2; ++which) {
        uint64_t& off = which ? P.ref_off[bp] : P.cons_off[bp];
        const int32_t len = which ? P.ref_len[bp] : P.cons_len[bp];
        if (pos + (uint64_t)len > cap || (len && !out_blob)) return fail(DELLYHIP_E_ARG, "out_blob too small for the probes");
        if (len) memcpy(out_blob + pos, h_blob.data() + off, (size_t)len);
        off = pos;
        pos += (uint64_t)len;
      }
    }
  }
  *used = pos;
  return 0;
}

int dellyhip_generate_probes_batch(dellyhip_ctx* c, int32_t n, const dellyhip_junction* junc, const char* seq_blob,
                                   const uint64_t* seq_off, uint64_t n_seq, dellyhip_probes* probes, char* out_blob,
                                   uint64_t cap, uint64_t* used) {
  dellyhip_batch* b = nullptr;
  int rc = dellyhip_batch_upload(c, n, junc, seq_blob, seq_off, n_seq, 16, &b);
  if (rc) return rc;
  rc = dellyhip_batch_run(c, b, nullptr);
  if (!rc) rc = dellyhip_batch_probes(c, b, probes, out_blob, cap, used);
  dellyhip_batch_free(c, b);
  return rc;
}


int dellyhip_lcs(dellyhip_ctx* c, const char* s1, int32_t m, const char* s2, int32_t n, int32_t* out) {
  if (!c || !out) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  return dh::msa_single_lcs(c->stream, s1, m, s2, n, out) ? fail(DELLYHIP_E_RUNTIME, "lcs") : 0;
}

int dellyhip_gotoh(dellyhip_ctx* c, const char* a1, int32_t r1, int32_t m, const char* a2, int32_t r2, int32_t n,
                   char* align_out, int32_t cap, int32_t* len, int32_t* score) {
  if (!c || !len || !score) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  int rc = dh::msa_single_gotoh(c->stream, c->params, c->msa_tmax, a1, r1, m, a2, r2, n, align_out, cap, len, score);
  return rc ? fail(rc, "gotoh") : 0;
}

int dellyhip_msa(dellyhip_ctx* c, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off, char* cs,
                 int32_t cs_cap, int32_t* cs_len, int32_t* rows) {
  if (!c || !cs_len || !rows || n_reads < 0) return fail(DELLYHIP_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  int rc = dh::msa_single(c->stream, c->params, c->msa_tmax, n_reads, seq_blob, seq_off, cs, cs_cap, cs_len, rows);
  return rc ? fail(rc, "msa") : 0;
}

}  // extern "C"
