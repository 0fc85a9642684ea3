/*
 * dellyhip_dropin.h -- C++ glue between the reference's header templates (namespace torali) and the C-ABI of
 * include/dellyhip.h.  The five headers next to this one (msa.h, needle.h, gotoh.h, split.h, assemble_msa.h) re-declare
 * the reference's entry points with their own template signatures; every call builds a batch of one (or, for
 * torali::refineBatch, of all junctions of a chromosome) and runs it on the MI355X.  There is no CPU path: a missing
 * device, a HIP error or a junction beyond a kernel limit throws dellyhip_dropin::Error (the reference has no error
 * channel on this path; a silent skip would change the VCF).
 *
 * One dellyhip_ctx per host thread and parameter set (thread_local): the reference calls msa()/alignConsensus() from
 * its ThreadPool workers (src/shortpe.h:175-201), a context is single-threaded.  All of them are created with
 * dellyhip_create_shared against ONE process-wide root context, so the resident chromosomes exist once per GPU however
 * many threads and parameter sets there are (a chromosome is uploaded by whichever thread sees it first).
 * Device: $DELLYHIP_DEVICE (default 0).
 *
 * The per-call wrappers (msa, alignConsensus, ...) run a batch of ONE junction per call: they exist so that every call
 * site compiles and gives the reference's result, not to fill a GPU.  The ThreadPool loop itself is torali::refineBatch
 * (split.h), one call per chromosome.
 */
#ifndef DELLYHIP_DROPIN_H
#define DELLYHIP_DROPIN_H

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../dellyhip.h"

namespace dellyhip_dropin {

struct Error : std::runtime_error {
  int code;
  Error(int c, std::string const& what) : std::runtime_error("dellyhip " + std::to_string(c) + ": " + what), code(c) {}
};

inline void check(int rc) {
  if (rc != 0) throw Error(rc, dellyhip_last_error());
}

/* ---- duck-typed TConfig -> dellyhip_params (aliscore, minCliqueSize, flankQuality,
 * minimumFlankSize, indelsize, minConsWindow; a config that lacks a field keeps the `delly sr` default) ---- */
#define DELLYHIP_DROPIN_FIELD(name)                                                                         \
  template <typename T, typename = void> struct has_##name : std::false_type {};                            \
  template <typename T> struct has_##name<T, std::void_t<decltype(std::declval<T const&>().name)>> : std::true_type {};
DELLYHIP_DROPIN_FIELD(aliscore)
DELLYHIP_DROPIN_FIELD(minCliqueSize)
DELLYHIP_DROPIN_FIELD(flankQuality)
DELLYHIP_DROPIN_FIELD(minimumFlankSize)
DELLYHIP_DROPIN_FIELD(indelsize)
DELLYHIP_DROPIN_FIELD(minConsWindow)
#undef DELLYHIP_DROPIN_FIELD

template <typename TConfig>
inline dellyhip_params make_params(TConfig const& c, bool realign = false) {
  dellyhip_params p;
  dellyhip_default_params_sr(&p);
  if constexpr (has_aliscore<TConfig>::value) {
    p.match = (int32_t)c.aliscore.match;
    p.mismatch = (int32_t)c.aliscore.mismatch;
    p.gap_open = (int32_t)c.aliscore.go;
    p.gap_extend = (int32_t)c.aliscore.ge;
  }
  if constexpr (has_minCliqueSize<TConfig>::value) p.min_clique_size = (int32_t)c.minCliqueSize;
  if constexpr (has_flankQuality<TConfig>::value) p.flank_quality = (float)c.flankQuality;
  if constexpr (has_minimumFlankSize<TConfig>::value) p.minimum_flank_size = (int32_t)c.minimumFlankSize;
  if constexpr (has_indelsize<TConfig>::value) p.indelsize = (int32_t)c.indelsize;
  if constexpr (has_minConsWindow<TConfig>::value) p.min_cons_window = (int32_t)c.minConsWindow;
  p.reserved = realign ? 1 : 0;   /* alignConsensus(..., realign): src/split.h:644-646, :564-572 */
  return p;
}

/* ---- the process-wide root: owns the shared chromosome table ---- */
struct Root {
  std::mutex mu;
  dellyhip_ctx* ctx = nullptr;
  std::map<int32_t, std::pair<const char*, int64_t>> chr;   /* chromosome index -> the host buffer that is resident */
  ~Root() {
    if (ctx) dellyhip_destroy(ctx);
  }
};
inline Root& root() {
  static Root R;
  return R;
}

/* ---- one context per thread and parameter set, all sharing the root's chromosomes ---- */
struct Session {
  dellyhip_ctx* ctx = nullptr;
  dellyhip_params params{};
  ~Session() {
    if (ctx) dellyhip_destroy(ctx);
  }
  /* alignConsensus(c, hdr, seq, sndSeq, sv): `seq` is the caller's faidx buffer of chromosome `idx`
   * (src/shortpe.h:88); it is uploaded -- once per process, not per thread -- when the (pointer, length) pair changes
   * (replacing a chromosome waits for the kernels of every context that may still read the old copy) */
  void chromosome(int32_t idx, const char* seq, int64_t len) {
    Root& R = root();
    std::lock_guard<std::mutex> g(R.mu);
    auto it = R.chr.find(idx);
    if (it != R.chr.end() && it->second.first == seq && it->second.second == len) return;
    check(dellyhip_set_chromosome(R.ctx, idx, seq, len));
    R.chr[idx] = std::make_pair(seq, len);
  }
};

inline Session& session(dellyhip_params const& p) {
  thread_local std::vector<std::unique_ptr<Session>> cache;
  for (auto& s : cache)
    if (std::memcmp(&s->params, &p, sizeof p) == 0) return *s;
  std::unique_ptr<Session> s(new Session());
  s->params = p;
  Root& R = root();
  {
    std::lock_guard<std::mutex> g(R.mu);
    if (!R.ctx) {
      const char* dev = std::getenv("DELLYHIP_DEVICE");
      check(dellyhip_create(&p, dev ? std::atoi(dev) : 0, &R.ctx));
    }
    check(dellyhip_create_shared(R.ctx, &p, &s->ctx));
  }
  cache.push_back(std::move(s));
  return *cache.back();
}

/* reverseComplement of src/util.h:549-563 as the path observes it: the reversed string is upper-cased, A/C/G/T/N are
 * complemented, and a position whose reversed letter is anything else KEEPS ITS ORIGINAL (un-reversed) byte */
inline void reverse_complement(std::string& s) {
  const std::size_t n = s.size();
  std::string out(s);
  for (std::size_t i = 0; i < n; ++i) {
    unsigned char ch = (unsigned char)s[n - 1 - i];
    if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
    switch (ch) {
      case 'A': out[i] = 'T'; break;
      case 'C': out[i] = 'G'; break;
      case 'G': out[i] = 'C'; break;
      case 'T': out[i] = 'A'; break;
      case 'N': out[i] = 'N'; break;
      default: break;
    }
  }
  s.swap(out);
}

/* reads of a set, concatenated in ITERATION ORDER (std::unordered_set order is part of the semantics) */
template <typename TSplitReadSet>
inline void pack_reads(TSplitReadSet const& sps, std::string& blob, std::vector<uint64_t>& off) {
  for (auto const& s : sps) {
    blob.append(s.data(), s.size());
    off.push_back((uint64_t)blob.size());
  }
}

}  // namespace dellyhip_dropin

#endif
