// comm.hpp -- multi-GPU return path of the refinement results: junctions shard across the GPUs of a
// node with no data-path collective; what rank 0 needs for mergeSort / VCF emission (src/delly.h:149,179) is every
// rank's fixed-size result records PLUS the variable-length consensus / "REF,ALT" bytes (src/split.h:606-637).
// RCCL has no gatherv: ranks exchange their (record count, blob bytes) with one ncclAllGather, then the non-root ranks
// ncclSend and the root ncclRecv both pieces inside one group (point-to-point over xGMI; tens of KB to a few MB per
// rank, latency-bound).  RCCL is loaded with dlopen at the first use, so a single-GPU run never touches it and the
// library has no link-time dependency on it.
//
// The gather protocol (dellyhip.hip: gather_device) is written against `Link`, the three operations it needs from a
// transport: an all-gather of two 64-bit words per rank, and a group of sends / receives of device buffers.  Two
// transports exist: `RcclLink` (one process per GPU, xGMI) and `HostLink` -- POSIX shared memory between the
// processes of one node, device buffers staged through the sender's pinned outbox.  RCCL refuses a communicator whose
// ranks share a device, so HostLink is what carries the SAME protocol code when several ranks are on one GPU
// (`bench.py --oversubscribe`, the two-process tests on a one-GPU box) and what the host-only tests of the abort
// protocol run on (no device needed for the size exchanges).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved with dlsym
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace dh {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

inline RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // an RCCL the process already holds (e.g. the one PyTorch ships) wins: two copies would each build their own
    // topology over the same xGMI links
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) { api.error = std::string("dlopen(librccl): ") + dlerror(); return; }
    auto sym = [&](const char* name) { void* p = dlsym(api.lib, name); if (!p && api.error.empty()) api.error = std::string("dlsym ") + name; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return api;
}

// ---- transports ------------------------------------------------------------------------------------------------------
// Every call returns 0 or a negative DELLYHIP_E_* value with the message in `err`.  allgather2 is synchronous (the
// words are on the host when it returns); the send / recv group has completed on `s` when group_end returns.
struct Link {
  int rank = 0, world = 1;
  std::string err;
  virtual ~Link() {}
  virtual const char* kind() const = 0;
  virtual int transport_ranks() = 0;   // the size the transport itself reports (ncclCommCount / attached processes)
  virtual int allgather2(const uint64_t mine[2], uint64_t* all, hipStream_t s) = 0;
  virtual int group_begin() = 0;
  virtual int send(const void* dev, uint64_t bytes, int peer, hipStream_t s) = 0;
  virtual int recv(void* dev, uint64_t bytes, int peer, hipStream_t s) = 0;
  virtual int group_end(hipStream_t s) = 0;
};

struct RcclLink final : Link {
  ncclComm_t nccl = nullptr;
  uint64_t* d_words = nullptr;   // 2 * world gathered words + 2 of this rank
  ncclResult_t pending = ncclSuccess;

  int fail_nccl(ncclResult_t r) { err = rccl_api().GetErrorString(r); return -3; }
  int fail_hip(hipError_t e) { err = std::string("HIP: ") + hipGetErrorString(e); return -3; }

  int init(const void* id128, int rank_, int world_) {
    rank = rank_;
    world = world_;
    RcclApi& A = rccl_api();
    if (!A.error.empty()) { err = A.error; return -3; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const ncclResult_t r = A.CommInitRank(&nccl, world, id, rank);
    if (r != ncclSuccess) return fail_nccl(r);
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_words), (2 * (size_t)world + 2) * sizeof(uint64_t));
    if (e != hipSuccess) { err = "hipMalloc of the communicator's exchange words"; return -5; }
    return 0;
  }
  ~RcclLink() override {
    if (nccl) (void)rccl_api().CommDestroy(nccl);
    if (d_words) (void)hipFree(d_words);
  }
  const char* kind() const override { return "rccl"; }
  int transport_ranks() override {
    int n = -1;
    RcclApi& A = rccl_api();
    if (!nccl || !A.CommCount || A.CommCount(nccl, &n) != ncclSuccess) return -1;
    return n;
  }
  int allgather2(const uint64_t mine[2], uint64_t* all, hipStream_t s) override {
    RcclApi& A = rccl_api();
    hipError_t e = hipMemcpyAsync(d_words + 2 * world, mine, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return fail_hip(e);
    const ncclResult_t r = A.AllGather(d_words + 2 * world, d_words, 2, ncclUint64, nccl, s);
    if (r != ncclSuccess) return fail_nccl(r);
    e = hipMemcpyAsync(all, d_words, 2 * (size_t)world * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return fail_hip(e);
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail_hip(e);
    return 0;
  }
  int group_begin() override {
    pending = rccl_api().GroupStart();
    return pending == ncclSuccess ? 0 : fail_nccl(pending);
  }
  int send(const void* dev, uint64_t bytes, int peer, hipStream_t s) override {
    if (pending == ncclSuccess && bytes) pending = rccl_api().Send(dev, (size_t)bytes, ncclUint8, peer, nccl, s);
    return 0;   // (errors surface in group_end: a started group must always be closed)
  }
  int recv(void* dev, uint64_t bytes, int peer, hipStream_t s) override {
    if (pending == ncclSuccess && bytes) pending = rccl_api().Recv(dev, (size_t)bytes, ncclUint8, peer, nccl, s);
    return 0;
  }
  int group_end(hipStream_t s) override {
    const ncclResult_t r2 = rccl_api().GroupEnd();
    if (pending != ncclSuccess) return fail_nccl(pending);
    if (r2 != ncclSuccess) return fail_nccl(r2);
    const hipError_t e = hipStreamSynchronize(s);
    return e == hipSuccess ? 0 : fail_hip(e);
  }
};

// HostLink: the processes of one node meet in a POSIX shared-memory control segment "/dellyhip_<name>_ctl" (rank 0
// creates it).  allgather2: every rank publishes its two words in the buffer of the round's parity and raises its
// round counter; a rank can be at most one round ahead of the slowest, so two buffers suffice.  A send group packs its
// parts (device -> the sender's outbox segment "/dellyhip_<name>_box<rank>_<generation>", grown by starting a new
// generation) and publishes (message number, generation, part sizes, destination); the receiver maps the outbox, copies
// the parts to its device buffers and acknowledges; the sender reuses the outbox only after the acknowledgement.
// Every wait has a deadline (DELLYHIP_LINK_TIMEOUT_S, default 120): a dead peer gives an error, not a hang.
struct HostLink final : Link {
  static constexpr uint64_t MAGIC = 0x64656c6c79686c31ull;   // "dellyhl1"
  static constexpr int MAX_PARTS = 8;
  static constexpr int MAX_RANKS = 64;
  // The message sequence is per (sender, destination): a receiver compares msg_to[its rank] with what IT has consumed.  (One
  // counter per sender -- round 4 -- broke as soon as two collectives had different roots: the second root found the
  // sender's counter ahead of its own count and took the message meant for the first root for its own.)
  struct Slot {
    std::atomic<uint64_t> attached;      // 1 once the rank has mapped the segment
    std::atomic<uint64_t> round;         // all-gather rounds this rank has published
    uint64_t words[2][2];                // [parity][word]
    std::atomic<uint64_t> msg_to[MAX_RANKS];     // messages this rank has published for each destination
    std::atomic<uint64_t> ack_from[MAX_RANKS];   // ... and how many of them that destination has consumed
    uint64_t gen, n_parts, part[MAX_PARTS];      // the message in the outbox (one at a time: rewritten only after its acknowledgement)
    char pad_[64];
  };
  struct Ctl {
    std::atomic<uint64_t> magic;
    uint64_t world;
    Slot slot[1];
  };
  struct Map {
    void* p = nullptr;
    size_t bytes = 0;
    bool pinned = false;
    void drop() {
      if (p) {
        if (pinned) (void)hipHostUnregister(p);
        munmap(p, bytes);
      }
      p = nullptr; bytes = 0; pinned = false;
    }
  };
  std::string name;
  Ctl* ctl = nullptr;
  size_t ctl_bytes = 0;
  uint64_t round = 0;
  Map box;                       // my outbox
  uint64_t box_gen = 0;
  std::vector<Map> peer_box;     // mapped outboxes of the peers
  std::vector<uint64_t> peer_gen, seen_msg;
  int last_dst = -1;             // destination of the message that occupies my outbox
  bool use_device = true;
  double timeout_s = 120.0;
  struct Part { void* dev; uint64_t bytes; int peer; bool is_send; };
  std::vector<Part> parts;

  int fail(int rc, const std::string& m) { err = "hostlink: " + m; return rc; }
  std::string seg(const char* what, int r = -1, uint64_t gen = 0) const {
    char b[160];
    if (r < 0) snprintf(b, sizeof b, "/dellyhip_%s_%s", name.c_str(), what);
    else snprintf(b, sizeof b, "/dellyhip_%s_%s%d_%llu", name.c_str(), what, r, (unsigned long long)gen);
    return b;
  }
  template <class F>
  bool wait_for(F&& ok) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      if (ok()) return true;
      if (spin < 2000) sched_yield();
      else usleep(50);
      if ((spin & 255) == 255) {
        follow_replacement();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
      }
    }
  }
  // a leftover control segment of a run that crashed BEFORE this rank attached (rank 0 attached, MAGIC set, this
  // rank's slot untouched) passes the freshness test of init().  Rank 0 of the new run unlinks it and creates another one of the
  // same name, so a rank that has not exchanged anything yet looks the name up again while it waits: a different inode is the
  // live segment, and the rank moves over (re-publishing the words of a first size exchange it has already posted).
  ino_t ctl_ino = 0;
  uint64_t last_words[2] = {0, 0};
  bool in_allgather = false;
  void follow_replacement() {
    if (rank == 0 || !ctl || last_dst >= 0 || !(round == 0 || (round == 1 && in_allgather))) return;
    for (uint64_t v : seen_msg)
      if (v) return;
    const std::string cn = seg("ctl");
    const int fd = shm_open(cn.c_str(), O_RDWR, 0600);
    if (fd < 0) return;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_ino == ctl_ino || (size_t)st.st_size < ctl_bytes) { close(fd); return; }
    void* p = mmap(nullptr, ctl_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return;
    Ctl* c = static_cast<Ctl*>(p);
    if (c->magic.load(std::memory_order_acquire) != MAGIC || c->world != (uint64_t)world) { munmap(p, ctl_bytes); return; }   // (not set up yet: next time)
    munmap(ctl, ctl_bytes);
    ctl = c;
    ctl_ino = st.st_ino;
    Slot& me = ctl->slot[rank];
    me.attached.store(1, std::memory_order_release);
    if (round == 1) {
      me.words[1][0] = last_words[0];
      me.words[1][1] = last_words[1];
      me.round.store(1, std::memory_order_release);
    }
  }

  int init(const char* name_, int rank_, int world_, bool device) {
    rank = rank_;
    world = world_;
    name = name_ ? name_ : "";
    use_device = device;
    if (name.empty() || name.size() > 96 || name.find('/') != std::string::npos) return fail(-2, "bad segment name");
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return fail(-2, "hostlink: 1 .. 64 ranks");
    if (const char* t = getenv("DELLYHIP_LINK_TIMEOUT_S")) timeout_s = std::max(0.05, atof(t));
    ctl_bytes = sizeof(Ctl) + (size_t)(world - 1) * sizeof(Slot);
    const std::string cn = seg("ctl");
    if (rank == 0) {
      shm_unlink(cn.c_str());   // (a stale segment of a crashed run)
      const int fd = shm_open(cn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)ctl_bytes) != 0) { if (fd >= 0) close(fd); return fail(-3, "cannot create " + cn); }
      void* p = mmap(nullptr, ctl_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) return fail(-3, "mmap of " + cn);
      ctl = static_cast<Ctl*>(p);
      ctl->world = (uint64_t)world;
      ctl->magic.store(MAGIC, std::memory_order_release);
    } else {
      // A segment of this name may be the leftover of a crashed run that rank 0 has not replaced yet: a fresh one has this
      // rank's slot untouched (ftruncate zero-fills), a leftover in which this rank took part has not.  Anything else is
      // dropped and looked up again until the deadline.
      const bool ok = wait_for([&] {
        const int fd = shm_open(cn.c_str(), O_RDWR, 0600);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < ctl_bytes) { close(fd); return false; }
        void* p = mmap(nullptr, ctl_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        Ctl* c = static_cast<Ctl*>(p);
        if (c->magic.load(std::memory_order_acquire) == MAGIC && c->slot[rank].attached.load(std::memory_order_acquire) == 0 &&
            c->slot[rank].round.load(std::memory_order_relaxed) == 0 && c->slot[0].attached.load(std::memory_order_acquire) != 2) {
          ctl = c;
          ctl_ino = st.st_ino;
          return true;
        }
        munmap(p, ctl_bytes);
        return false;
      });
      if (!ok) return fail(-3, "rank 0 never created (a fresh) " + cn);
    }
    if (ctl->world != (uint64_t)world) return fail(-2, "ranks disagree about the world size");
    ctl->slot[rank].attached.store(1, std::memory_order_release);
    peer_box.resize(world);
    peer_gen.assign(world, 0);
    seen_msg.assign(world, 0);
    return 0;
  }
  bool outbox_free() const {   // the message in my outbox (if any) has been consumed by its destination
    if (last_dst < 0) return true;
    const Slot& me = ctl->slot[rank];
    return me.ack_from[last_dst].load(std::memory_order_acquire) >= me.msg_to[last_dst].load(std::memory_order_relaxed);
  }
  ~HostLink() override {
    if (ctl && box.p)   // a receiver may still be reading the last message
      (void)wait_for([&] { return outbox_free(); });
    if (box.p) { box.drop(); shm_unlink(seg("box", rank, box_gen).c_str()); }
    for (Map& m : peer_box) m.drop();
    if (ctl) {
      ctl->slot[rank].attached.store(2, std::memory_order_release);
      if (rank == 0) {   // the creator unlinks once everybody else has left (or the deadline passes)
        (void)wait_for([&] { for (int r = 1; r < world; ++r) if (ctl->slot[r].attached.load(std::memory_order_acquire) == 1) return false; return true; });
        shm_unlink(seg("ctl").c_str());
      }
      munmap(ctl, ctl_bytes);
    }
  }
  const char* kind() const override { return "hostlink"; }
  int transport_ranks() override {
    int n = 0;
    for (int r = 0; r < world; ++r) n += ctl->slot[r].attached.load(std::memory_order_acquire) != 0;
    return n;
  }
  int allgather2(const uint64_t mine[2], uint64_t* all, hipStream_t) override {
    const uint64_t k = ++round;
    Slot& me = ctl->slot[rank];
    me.words[k & 1][0] = mine[0];
    me.words[k & 1][1] = mine[1];
    last_words[0] = mine[0];
    last_words[1] = mine[1];
    me.round.store(k, std::memory_order_release);
    for (int r = 0; r < world; ++r) {
      in_allgather = true;    // (the wait may move this rank to the segment that replaced a leftover: ctl is looked up in the loop)
      const bool got = wait_for([&] { return ctl->slot[r].round.load(std::memory_order_acquire) >= k; });
      in_allgather = false;
      if (!got) {
        char b[96];
        snprintf(b, sizeof b, "timed out waiting for rank %d in the size exchange", r);
        return fail(-3, b);
      }
      Slot& s = ctl->slot[r];
      all[2 * r] = s.words[k & 1][0];
      all[2 * r + 1] = s.words[k & 1][1];
    }
    return 0;
  }
  int group_begin() override { parts.clear(); return 0; }
  int send(const void* dev, uint64_t bytes, int peer, hipStream_t) override {
    if (bytes) parts.push_back({const_cast<void*>(dev), bytes, peer, true});
    return 0;
  }
  int recv(void* dev, uint64_t bytes, int peer, hipStream_t) override {
    if (bytes) parts.push_back({dev, bytes, peer, false});
    return 0;
  }
  int copy(void* dst, const void* src, uint64_t bytes, hipMemcpyKind k, hipStream_t s) {
    if (!use_device) { memcpy(dst, src, bytes); return 0; }   // (host-only tests: "device" pointers are host pointers)
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, k, s);
    return e == hipSuccess ? 0 : fail(-3, std::string("HIP: ") + hipGetErrorString(e));
  }
  int ensure_box(uint64_t need) {
    if (box.p && box.bytes >= need) return 0;
    if (box.p) { box.drop(); shm_unlink(seg("box", rank, box_gen).c_str()); }
    ++box_gen;
    const size_t cap = (size_t)std::max<uint64_t>(need + need / 4, 1u << 20);
    const std::string bn = seg("box", rank, box_gen);
    shm_unlink(bn.c_str());
    const int fd = shm_open(bn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)cap) != 0) { if (fd >= 0) close(fd); return fail(-5, "cannot create " + bn); }
    void* p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(-5, "mmap of " + bn);
    box.p = p;
    box.bytes = cap;
    box.pinned = use_device && hipHostRegister(p, cap, hipHostRegisterDefault) == hipSuccess;
    if (use_device && !box.pinned) (void)hipGetLastError();
    return 0;
  }
  int group_end(hipStream_t s) override {
    // sends first (one message to one destination per group), then the receives in the order they were posted
    uint64_t total = 0, n_send = 0;
    int dst = -1;
    for (const Part& p : parts)
      if (p.is_send) {
        if (dst >= 0 && dst != p.peer) return fail(-2, "one destination per send group");
        dst = p.peer;
        total += p.bytes;
        ++n_send;
      }
    if (n_send > MAX_PARTS) return fail(-2, "too many parts in one send group");
    if (n_send) {
      Slot& me = ctl->slot[rank];
      if (dst < 0 || dst >= world) return fail(-2, "send to a rank outside the communicator");
      if (!wait_for([&] { return outbox_free(); }))
        return fail(-3, "timed out waiting for the previous message to be consumed");
      if (int rc = ensure_box(total)) return rc;
      uint64_t at = 0, k = 0;
      for (const Part& p : parts)
        if (p.is_send) {
          if (int rc = copy(static_cast<char*>(box.p) + at, p.dev, p.bytes, hipMemcpyDeviceToHost, s)) return rc;
          me.part[k++] = p.bytes;
          at += p.bytes;
        }
      if (use_device) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(-3, std::string("HIP: ") + hipGetErrorString(e));
      }
      me.gen = box_gen;
      me.n_parts = n_send;
      last_dst = dst;
      me.msg_to[dst].store(me.msg_to[dst].load(std::memory_order_relaxed) + 1, std::memory_order_release);
    }
    for (size_t i = 0; i < parts.size();) {
      if (parts[i].is_send) { ++i; continue; }
      const int q = parts[i].peer;
      Slot& sq = ctl->slot[q];
      if (q < 0 || q >= world) return fail(-2, "receive from a rank outside the communicator");
      if (!wait_for([&] { return sq.msg_to[rank].load(std::memory_order_acquire) > seen_msg[q]; })) {
        char b[96];
        snprintf(b, sizeof b, "timed out waiting for the payload of rank %d", q);
        return fail(-3, b);
      }
      if (peer_gen[q] != sq.gen || !peer_box[q].p) {
        peer_box[q].drop();
        const std::string bn = seg("box", q, sq.gen);
        const int fd = shm_open(bn.c_str(), O_RDWR, 0600);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return fail(-3, "cannot open " + bn); }
        void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return fail(-3, "mmap of " + bn);
        peer_box[q].p = p;
        peer_box[q].bytes = (size_t)st.st_size;
        peer_box[q].pinned = use_device && hipHostRegister(p, peer_box[q].bytes, hipHostRegisterDefault) == hipSuccess;
        if (use_device && !peer_box[q].pinned) (void)hipGetLastError();
        peer_gen[q] = sq.gen;
      }
      uint64_t at = 0, k = 0;
      for (; i < parts.size() && !parts[i].is_send && parts[i].peer == q; ++i, ++k) {
        if (k >= sq.n_parts || sq.part[k] != parts[i].bytes) return fail(-3, "a receive does not match the peer's send");
        if (int rc = copy(parts[i].dev, static_cast<char*>(peer_box[q].p) + at, parts[i].bytes, hipMemcpyHostToDevice, s)) return rc;
        at += parts[i].bytes;
      }
      if (k != sq.n_parts) return fail(-3, "a peer sent more parts than were received");
      if (use_device) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(-3, std::string("HIP: ") + hipGetErrorString(e));
      }
      seen_msg[q] = sq.msg_to[rank].load(std::memory_order_relaxed);
      sq.ack_from[rank].store(seen_msg[q], std::memory_order_release);
    }
    parts.clear();
    return 0;
  }
};

// Cost-balanced assignment of junctions to ranks: predicted cost N^2 L^2 (all-pairs LCS / NW) +
// (N-1) L^2 c2 (progressive alignment) + m n (split alignment).  Longest-processing-time-first greedy: junctions by
// decreasing cost, each to the currently lightest rank.  owner[i] = rank of junction i; deterministic (ties by index).
inline void balance_by_cost(const double* cost, int n, int world, int32_t* owner) {
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  std::vector<double> load(std::max(world, 1), 0.0);
  for (int i : order) {
    int best = 0;
    for (int r = 1; r < world; ++r)
      if (load[r] < load[best]) best = r;
    owner[i] = best;
    load[best] += cost[i];
  }
}

}  // namespace dh
