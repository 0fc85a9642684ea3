// msa_kernel.hpp -- gfx950 device code for msa() (src/msa.h:185-239), one
// junction per 64-lane wavefront:
//   distanceMatrix/lcs   src/msa.h:10-44   -> bit-parallel LCS, one read pair per lane
//   upgma                src/msa.h:46-89   -> lane-parallel arg-max with the reference's
//                                             first-row-major tie-break, matrix in LDS (int8)
//   palign/gotoh         src/msa.h:91-109, src/gotoh.h:71-174, src/align.h:89-229
//                                          -> anti-diagonal affine DP (rows = columns of a1
//                                             owned by lanes, DPP hand-off), profile score in
//                                             the reference's float evaluation order, 4 trace
//                                             bits per cell in global scratch
//   consensus            src/msa.h:111-173 -> column-parallel vote
//
// Float semantics: profile entries are count/sum float divisions; a cell's
// score is sum_{k1<5} sum_{k2<5} (p1[k1]*p2[k2])*w(k1,k2) accumulated in float in that order
// and truncated to int.  Terms with p1[k1]==0 or p2[k2]==0 are exactly +-0 and x + (+-0) == x,
// so only the non-zero entries are visited (in the same order).  Built with
// -ffp-contract=off: no FMA contraction, like the reference's x86-64 build.
#pragma once
#include "dev_pool.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/dellyhip.h"
#include "split_kernel.hpp"
#include "split_pk.hpp"   // pk, pk_add / pk_sub / pk_max: the packed Gotoh pass of msa_body.inc


// MSA_SYNC(): the hand-over between the lanes of ONE wavefront (LDS / workspace written by some lanes, read by others).  The
// kernels of this file run one wavefront per workgroup, where __syncthreads() is exactly this fence (the compiler drops the
// s_barrier of a single-wave workgroup); the team kernel runs several wavefronts of a workgroup through DIFFERENT merges at
// the same time, where a real barrier inside a merge would deadlock.
#ifndef MSA_SYNC
#define MSA_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#endif

namespace dh {

// shared by both instances of the MSA code (msa_body.inc)
constexpr int GKMAX = 8;     // Gotoh rows per lane in one pass; longer nodes run in strips of 64*GKMAX rows (msa_big)
constexpr int GINF = 1000000;  // DnaScore::inf, src/align.h:21
constexpr int PROFW = 8;     // dwords per profile column: meta + 5 values (+2 pad)
constexpr int TMAXC = 96;    // profile column types per alignment node the score table is built for ...
constexpr int TABCAP = 4096; // ... with T1 x T2 <= TABCAP table entries (64 x 64, 96 x 42, ...): LDS per wavefront decides the occupancy
constexpr int HSLOTS = 128;  // open-addressing table of column-type keys
constexpr int FASTK = 5;      // rows per lane served by the score-table kernel (node length <= 319)
constexpr int MSA_DEFER = 1; // merge_nodes: more column types than the table holds -> direct-float kernel

struct MsaArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  dellyhip_params p;
  dellyhip_result* res;
  uint8_t* out_blob;      // consensus goes to out_blob + j*out_stride
  uint64_t out_stride;
  int32_t* cons_len;
  uint8_t* ws;            // per resident block workspace
  uint64_t ws_stride;
  int32_t n_work;
  int32_t* work_counter;
  int32_t tmax;           // column types per node handled by the score table (<= TMAXC)
  int32_t* defer_counter; // junctions handed to the direct-float kernel
  int32_t* big_counter;   // junctions beyond the limits of the standard instance (-> msa_big kernel); may be null
  int32_t out_cons_cap;   // bytes of the consensus slot at out_blob + j*out_stride
  const int32_t* order;   // the junction the w-th claim of the work counter gets (most expensive first), or null: junction w
  int32_t pair;           // 1: two merges of a junction share a Gotoh pass where they fit (gotoh_pass_pair); 0: one merge per pass
  // single-item gotoh mode (dellyhip_gotoh): two given alignments
  const uint8_t* g_a1;
  const uint8_t* g_a2;
  int32_t g_r1, g_m, g_r2, g_n;
  uint8_t* g_out;         // (r1+r2) x LCAP
  int32_t* g_info;        // [0]=len, [1]=score, [2]=status
};

// ---- K1: bit-parallel LCS (Crochemore et al. / Hyyro): V' = (V + (V & M)) | (V & ~M) ----
__device__ __forceinline__ int letter_code(uint8_t c) { return letter_code_bf(c); }   // A 0, C 1, G 2, T 3, N 4, anything else -1


// ---- alignment node descriptor ------------------------------------------------
struct Node {
  const uint8_t* p;  // row-major chars
  int rows, len, stride;
};


// (int) score as x86-64 cvttss2si does it: NaN / out of range -> 0x80000000
__device__ __forceinline__ int cvt_x86(float f) {
  if (!(f == f) || f >= 2147483648.0f || f < -2147483648.0f) return (int)0x80000000;
  return (int)f;
}


// ---- K7: consensus  src/msa.h:111-173.  Writes the ungapped consensus (<= cap bytes), returns its length.
template <typename LT>
__device__ __forceinline__ int consensus_node(const Node& a, const dellyhip_params& P, uint8_t* cs, int cap, LT& L,
                                              int lane) {
  for (int i = 0; i < a.rows; ++i) {
    int first = a.len, last = -1;
    for (int base = 0; base < a.len; base += WAVE) {
      int j = base + lane;
      bool nz = (j < a.len) && (a.p[(size_t)i * a.stride + j] != '-');
      unsigned long long bm = __ballot(nz);
      if (bm) {
        if (last == -1) first = base + __builtin_ctzll(bm);
        last = base + 63 - __builtin_clzll(bm);
      }
    }
    if (lane == 0) {
      L.first[i] = first;
      L.last[i] = last;
    }
  }
  MSA_SYNC();
  const int thr = max(2, min(P.min_clique_size, a.rows));
  int outn = 0;
  for (int base = 0; base < a.len; base += WAVE) {
    int j = base + lane;
    uint8_t letter = 0;
    if (j < a.len) {
      int cov = 0, cnt[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < a.rows; ++i) {
        if (L.first[i] <= j && j <= L.last[i]) {
          ++cov;
          uint8_t ch = a.p[(size_t)i * a.stride + j];
          if (ch == 'A' || ch == 'a') ++cnt[0];
          else if (ch == 'C' || ch == 'c') ++cnt[1];
          else if (ch == 'G' || ch == 'g') ++cnt[2];
          else if (ch == 'T' || ch == 't') ++cnt[3];
          else ++cnt[4];
        }
      }
      if (cov >= thr) {
        int mi = 0, mc = cnt[0];
#pragma unroll
        for (int q = 1; q < 5; ++q)
          if (cnt[q] > mc) {
            mc = cnt[q];
            mi = q;
          }
        if (mi < 4) letter = (uint8_t)("ACGT"[mi]);
      }
    }
    unsigned long long km = __ballot(letter != 0);
    unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int pos = outn + __popcll(km & below);
    if (letter && pos < cap) cs[pos] = letter;
    outn += __popcll(km);
  }
  return outn;
}


// consensus (src/msa.h:111-173) from a node's column statistics (msa_body.inc): cover count, A / C / G / T counts, and the
// bucket of everything else = cover - (A + C + G + T)
__device__ __forceinline__ int consensus_stats(const unsigned long long* stats, int len, int rows, const dellyhip_params& P, uint8_t* cs,
                                               int cap, int lane) {
  const int thr = max(2, min(P.min_clique_size, rows));
  int outn = 0;
  for (int base = 0; base < len; base += WAVE) {
    const int j = base + lane;
    uint8_t letter = 0;
    if (j < len) {
      const unsigned long long st = stats[j];
      const int cov = (int)((st >> 48) & 0xff);
      int cnt[5];
#pragma unroll
      for (int q = 0; q < 4; ++q) cnt[q] = (int)((st >> (8 * q)) & 0xff);
      cnt[4] = cov - (cnt[0] + cnt[1] + cnt[2] + cnt[3]);
      if (cov >= thr) {
        int mi = 0, mc = cnt[0];
#pragma unroll
        for (int q = 1; q < 5; ++q)
          if (cnt[q] > mc) {
            mc = cnt[q];
            mi = q;
          }
        if (mi < 4) letter = (uint8_t)("ACGT"[mi]);
      }
    }
    const unsigned long long km = __ballot(letter != 0);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int pos = outn + __popcll(km & below);
    if (letter && pos < cap) cs[pos] = letter;
    outn += __popcll(km);
  }
  return outn;
}

}  // namespace dh

#define DH_MSA_NS msa_std
#define DH_MSA_NRMAX 24
#define DH_MSA_RLMAX 256
#define DH_MSA_LCAP 512
#define DH_MSA_FAST 1
#define DH_MSA_LCS_GLOBAL 0
#include "msa_body.inc"
#undef DH_MSA_NS
#undef DH_MSA_NRMAX
#undef DH_MSA_RLMAX
#undef DH_MSA_LCAP
#undef DH_MSA_FAST
#undef DH_MSA_LCS_GLOBAL

#define DH_MSA_NS msa_big
#define DH_MSA_NRMAX 64
#define DH_MSA_RLMAX 1024
#define DH_MSA_LCAP 2048
#define DH_MSA_FAST 0
#define DH_MSA_LCS_GLOBAL 1
#include "msa_body.inc"
#undef DH_MSA_NS
#undef DH_MSA_NRMAX
#undef DH_MSA_RLMAX
#undef DH_MSA_LCAP
#undef DH_MSA_FAST
#undef DH_MSA_LCS_GLOBAL

namespace dh {
using namespace msa_std;   // NRMAX, LCAP, MsaWs ... of the standard instance are the unqualified names

// ---- host helpers -----------------------------------------------------------------
// What a batch needs from the two instances.  nmax: reads of the largest junction the standard instance takes;
// big_*: junctions that are certain (more reads / longer reads than msa_std holds) or likely (reads whose alignment can
// exceed its 512 columns) to be re-done by msa_big; maxlen: longest read of the batch.
struct MsaPlan {
  int nmax = 2;
  uint64_t ws_stride = 0;
  int big_nmax = 2;
  int big_count = 0;
  uint64_t big_ws_stride = 0;
  int maxlen = 1;
};
inline MsaPlan msa_prepare(const std::vector<dellyhip_junction>& junc, const uint64_t* seq_off) {
  MsaPlan P;
  for (auto const& J : junc) {
    if (J.n_seq < 2) continue;
    int jl = 1;
    for (int k = 0; k < J.n_seq; ++k)
      jl = std::max<int>(jl, (int)std::min<uint64_t>(seq_off[J.seq_first + k + 1] - seq_off[J.seq_first + k], 1u << 20));
    P.maxlen = std::max(P.maxlen, jl);
    const bool big = J.n_seq > NRMAX || jl > RLMAX || 2 * jl > LCAP - 8;
    if (big) {
      ++P.big_count;
      P.big_nmax = std::max(P.big_nmax, std::min<int>(J.n_seq, msa_big::NRMAX));
    }
    if (J.n_seq <= NRMAX) P.nmax = std::max(P.nmax, J.n_seq);
  }
  P.big_nmax = std::max(P.big_nmax, P.nmax);   // (an unpredicted overflow of msa_std lands in msa_big with <= NRMAX reads)
  P.ws_stride = (MsaWs::bytes(P.nmax) + 255) & ~255ull;
  P.big_ws_stride = (msa_big::MsaWs::bytes(P.big_nmax) + 255) & ~255ull;
  return P;
}

// score-table types need |match|, |mismatch| <= 127 (int8 table); otherwise every profile merge
// goes to the direct-float kernel
inline int msa_tmax(const dellyhip_params& P, int wanted) {
  if (P.match > 127 || P.match < -127 || P.mismatch > 127 || P.mismatch < -127) return 0;
  return std::max(0, std::min(wanted, TMAXC));
}

// a.work_counter, a.defer_counter and a.big_counter must be zeroed on the stream before the call.  big_ws: workspace of
// msa_big (big_grid blocks of big_stride bytes); junctions the standard instance flags with DELLYHIP_E_LIMIT are re-done there.
// Wavefronts per junction of the score-table kernel (msa_team_kernel): one while a launch has enough junctions for every
// resident slot, two or four when it does not (DELLYHIP_MSA_TEAM overrides; `delly sr` hands over 10^3 .. 10^4 per chromosome)
inline int msa_team_waves(int n_junctions, int slots, int forced) {
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  // Round 5: one wavefront per junction now runs two merges per Gotoh pass (msa_body.inc, gotoh_pass_pair), the teams one:
  // measured on 20-read junctions (tools/msa_rate.py) 500 per launch 1.19 / 1.04 / 0.79 ms with 1 / 2 / 4 wavefronts per
  // junction, 1 000: 1.25 / 1.26 / 1.34, 2 000: 1.66 / 1.86 / 2.17 -- a team only pays while the chip is less than a fifth full.
  if ((long long)n_junctions * 16 <= (long long)slots * 3) return 4;
  return 1;
}
inline uint64_t msa_team_stride(int nmax, int team) { return team <= 1 ? ((MsaWs::bytes(nmax) + 255) & ~255ull) : ((MsaTeamWs::bytes(nmax, team) + 255) & ~255ull); }
// blocks of `team` wavefronts that keep `slots` wavefront slots busy
inline int msa_team_grid(int n_junctions, int slots, int team) { return std::max(1, std::min(n_junctions, slots / std::max(team, 1))); }

inline int msa_launch(const MsaArgs& a, int grid, int nmax, hipStream_t s, uint8_t* big_ws = nullptr, uint64_t big_stride = 0,
                      int big_grid = 0, int big_nmax = 0, int team = 1) {
  if (team == 4) hipLaunchKernelGGL(msa_team_kernel<4>, dim3(grid), dim3(WAVE * 4), 0, s, a, nmax);
  else if (team == 2) hipLaunchKernelGGL(msa_team_kernel<2>, dim3(grid), dim3(WAVE * 2), 0, s, a, nmax);
  else hipLaunchKernelGGL(msa_kernel, dim3(grid), dim3(WAVE), 0, s, a, nmax);
  hipLaunchKernelGGL(msa_slow_kernel, dim3(grid), dim3(WAVE), 0, s, a, nmax);
  if (big_ws && big_grid > 0) {
    MsaArgs b = a;
    b.ws = big_ws;
    b.ws_stride = big_stride;
    hipLaunchKernelGGL(msa_big::msa_big_kernel, dim3(big_grid), dim3(WAVE), 0, s, b, big_nmax);
  }
  return 0;
}

inline int msa_single_lcs(hipStream_t s, const char* s1, int m, const char* s2, int n, int32_t* out) {
  if (m < 0 || n < 0) return DELLYHIP_E_ARG;
  if (m > msa_big::RLMAX && n <= msa_big::RLMAX) { std::swap(s1, s2); std::swap(m, n); }   // (the LCS length is symmetric; rows = s1)
  if (m > msa_big::RLMAX) return DELLYHIP_E_LIMIT;
  uint8_t *d1 = nullptr, *d2 = nullptr;
  int* dout = nullptr;
  if (dev_alloc((void**)&d1, std::max(m, 1) + 8) != hipSuccess || dev_alloc((void**)&d2, std::max(n, 1) + 8) != hipSuccess ||   // (+8: quadword reads)
      dev_alloc((void**)&dout, 4) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(d1, s1, m, hipMemcpyHostToDevice);
  (void)hipMemcpy(d2, s2, n, hipMemcpyHostToDevice);
  if (m <= RLMAX) hipLaunchKernelGGL(lcs_single_kernel, dim3(1), dim3(WAVE), 0, s, d1, m, d2, n, dout);
  else hipLaunchKernelGGL(msa_big::lcs_single_kernel, dim3(1), dim3(WAVE), 0, s, d1, m, d2, n, dout);
  hipError_t e = hipStreamSynchronize(s);
  (void)hipMemcpy(out, dout, 4, hipMemcpyDeviceToHost);
  dev_free(d1);
  dev_free(d2);
  dev_free(dout);
  return e == hipSuccess ? 0 : DELLYHIP_E_RUNTIME;
}

inline int msa_single_gotoh(hipStream_t s, const dellyhip_params& P, int tmax, const char* a1, int r1, int m,
                            const char* a2, int r2, int n, char* out, int cap, int32_t* len, int32_t* score) {
  if (r1 < 1 || r2 < 1 || m < 0 || n < 0) return DELLYHIP_E_ARG;
  const bool big = r1 > NRMAX || r2 > NRMAX || m > LCAP - 1 || n > LCAP || m + n > LCAP;   // (output of m + n columns at most)
  if (big && (r1 > msa_big::NRMAX || r2 > msa_big::NRMAX || m > msa_big::LCAP - 1 || n > msa_big::LCAP)) return DELLYHIP_E_LIMIT;
  const int LC = big ? msa_big::LCAP : LCAP;
  uint8_t *d1 = nullptr, *d2 = nullptr, *dout = nullptr, *ws = nullptr;
  int* dinfo = nullptr;
  size_t wsb = big ? msa_big::MsaWs::bytes(2) : MsaWs::bytes(2);
  if (dev_alloc((void**)&d1, std::max(r1 * m, 1)) != hipSuccess || dev_alloc((void**)&d2, std::max(r2 * n, 1)) != hipSuccess ||
      dev_alloc((void**)&dout, (size_t)2 * (r1 + r2) * LC) != hipSuccess || dev_alloc((void**)&ws, wsb) != hipSuccess ||
      dev_alloc((void**)&dinfo, 32) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(d1, a1, (size_t)r1 * m, hipMemcpyHostToDevice);
  (void)hipMemcpy(d2, a2, (size_t)r2 * n, hipMemcpyHostToDevice);
  MsaArgs A{};
  A.p = P;
  A.ws = ws;
  A.g_a1 = d1; A.g_a2 = d2; A.g_r1 = r1; A.g_m = m; A.g_r2 = r2; A.g_n = n;
  A.g_out = dout;
  A.g_info = dinfo;
  A.tmax = msa_tmax(P, tmax);
  if (big) hipLaunchKernelGGL(msa_big::gotoh_single_big_kernel, dim3(1), dim3(WAVE), 0, s, A);
  else hipLaunchKernelGGL(gotoh_single_kernel, dim3(1), dim3(WAVE), 0, s, A);
  hipError_t e = hipStreamSynchronize(s);
  int info[3] = {0, 0, DELLYHIP_E_RUNTIME};
  if (e == hipSuccess) (void)hipMemcpy(info, dinfo, 12, hipMemcpyDeviceToHost);
  int rc = info[2];
  if (!rc) {
    *len = info[0];
    *score = info[1];
    if (info[0] > cap) rc = DELLYHIP_E_ARG;
    else {
      std::vector<uint8_t> tmp((size_t)(r1 + r2) * LC);
      (void)hipMemcpy(tmp.data(), dout, tmp.size(), hipMemcpyDeviceToHost);
      for (int i = 0; i < r1 + r2; ++i) memcpy(out + (size_t)i * cap, tmp.data() + (size_t)i * LC, info[0]);
    }
  }
  dev_free(d1); dev_free(d2); dev_free(dout); dev_free(ws); dev_free(dinfo);
  return rc;
}

// msa(c, sps, cs) for one read set (dellyhip_msa)
inline int msa_single(hipStream_t s, const dellyhip_params& P, int tmax, int n_reads, const char* seq_blob,
                      const uint64_t* seq_off, char* cs, int cs_cap, int32_t* cs_len, int32_t* rows) {
  if (n_reads > msa_big::NRMAX) return DELLYHIP_E_LIMIT;
  dellyhip_junction J{};
  J.n_seq = n_reads;
  J.seq_first = 0;
  uint64_t blob_bytes = n_reads ? seq_off[n_reads] : 0;
  const MsaPlan plan = msa_prepare(std::vector<dellyhip_junction>(1, J), seq_off);
  const int ccap = msa_big::LCAP;
  dellyhip_junction* dj = nullptr;
  uint8_t *dblob = nullptr, *dout = nullptr, *ws = nullptr, *wsb = nullptr;
  uint64_t* doff = nullptr;
  dellyhip_result* dres = nullptr;
  int32_t *dlen = nullptr, *dcnt = nullptr;
  if (dev_alloc((void**)&dj, sizeof J) != hipSuccess || dev_alloc((void**)&dblob, std::max<uint64_t>(blob_bytes, 1) + 64) != hipSuccess ||   // (+64: quadword reads)
      dev_alloc((void**)&doff, (n_reads + 1) * 8) != hipSuccess || dev_alloc((void**)&dres, sizeof(dellyhip_result)) != hipSuccess ||
      dev_alloc((void**)&dout, ccap) != hipSuccess || dev_alloc((void**)&ws, plan.ws_stride) != hipSuccess ||
      dev_alloc((void**)&wsb, plan.big_ws_stride) != hipSuccess || dev_alloc((void**)&dlen, 4) != hipSuccess ||
      dev_alloc((void**)&dcnt, 64) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(dj, &J, sizeof J, hipMemcpyHostToDevice);
  (void)hipMemcpy(dblob, seq_blob, blob_bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(doff, seq_off, (n_reads + 1) * 8, hipMemcpyHostToDevice);
  // (on the launch stream: hipMemset on the null stream is neither host-synchronous nor ordered with a non-blocking stream)
  (void)hipMemsetAsync(dres, 0, sizeof(dellyhip_result), s);
  (void)hipMemsetAsync(dcnt, 0, 64, s);
  (void)hipMemsetAsync(dlen, 0, 4, s);
  MsaArgs A{};
  A.junc = dj; A.seq_blob = dblob; A.seq_off = doff; A.p = P; A.res = dres; A.out_blob = dout; A.out_stride = ccap;
  A.out_cons_cap = ccap;
  A.cons_len = dlen; A.ws = ws; A.ws_stride = plan.ws_stride; A.n_work = 1; A.work_counter = dcnt;
  A.defer_counter = dcnt + 1;
  A.big_counter = dcnt + 2;
  A.tmax = msa_tmax(P, tmax);
  A.pair = 1;
  msa_launch(A, 1, plan.nmax, s, wsb, plan.big_ws_stride, 1, plan.big_nmax);
  hipError_t e = hipStreamSynchronize(s);
  int rc = (e == hipSuccess) ? 0 : DELLYHIP_E_RUNTIME;
  dellyhip_result R{};
  int32_t L = 0;
  if (!rc) {
    (void)hipMemcpy(&R, dres, sizeof R, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&L, dlen, 4, hipMemcpyDeviceToHost);
    if (R.status) rc = R.status;
    else {
      *rows = R.sr_support;
      *cs_len = L;
      if (L > cs_cap) rc = DELLYHIP_E_ARG;
      else if (L > 0) (void)hipMemcpy(cs, dout, L, hipMemcpyDeviceToHost);
    }
  }
  dev_free(dj); dev_free(dblob); dev_free(doff); dev_free(dres); dev_free(dout);
  dev_free(ws); dev_free(wsb); dev_free(dlen); dev_free(dcnt);
  return rc;
}

}  // namespace dh
