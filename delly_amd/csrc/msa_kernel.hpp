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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/dellyhip.h"
#include "split_kernel.hpp"

namespace dh {

constexpr int NRMAX = 32;    // reads per junction (delly sr default cap: 20, src/delly.h:224)
constexpr int RLMAX = 256;   // read length (bit-parallel LCS uses 4 x 64-bit words)
constexpr int LCSW = RLMAX / 64;
constexpr int LCAP = 512;    // alignment columns of any MSA node
constexpr int GKMAX = 8;     // Gotoh rows per lane: 64*8 >= LCAP
constexpr int NODES = 2 * NRMAX + 1;
constexpr int GINF = 1000000;  // DnaScore::inf, src/align.h:21
constexpr int PROFW = 8;     // dwords per profile column: meta + 5 values (+2 pad)

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
  // single-item gotoh mode (dellyhip_gotoh): two given alignments
  const uint8_t* g_a1;
  const uint8_t* g_a2;
  int32_t g_r1, g_m, g_r2, g_n;
  uint8_t* g_out;         // (r1+r2) x LCAP
  int32_t* g_info;        // [0]=len, [1]=score, [2]=status
};

// workspace layout per block
struct MsaWs {
  static __host__ __device__ uint64_t node_rows_cap(int nmax) { return (uint64_t)nmax * (nmax + 1) / 2 + 2; }
  static __host__ __device__ uint64_t bytes(int nmax) {
    uint64_t aln = node_rows_cap(nmax) * LCAP;                 // node alignments (chars)
    uint64_t prof = 2ull * LCAP * PROFW * 4;                   // two profiles
    uint64_t bits = ((uint64_t)(LCAP + 64 + 16) / 8 + 2) * GKMAX * WAVE * 4;  // trace nibbles
    return ((aln + 255) & ~255ull) + prof + bits;
  }
};

struct __attribute__((aligned(16))) MsaLds {
  unsigned long long lcsmask[NRMAX][5][LCSW];
  int8_t d[NODES * NODES];
  int16_t par[NODES], lch[NODES], rch[NODES];
  int32_t node_rows[NODES], node_len[NODES], node_base[NODES];
  uint32_t roff[NRMAX];   // read offsets relative to the junction's first read
  int32_t rlen[NRMAX];
  int32_t first[NRMAX], last[NRMAX];
  uint8_t trace[2 * LCAP + 8];
  uint8_t keep[LCAP];
};

// ---- K1: bit-parallel LCS (Crochemore et al. / Hyyro): V' = (V + (V & M)) | (V & ~M) ----
__device__ __forceinline__ int letter_code(uint8_t c) {
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : -1;
}

__device__ __forceinline__ int lcs_bitparallel(const unsigned long long (*maskI)[LCSW], const uint8_t* si, int li,
                                               const uint8_t* sj, int lj) {
  unsigned long long V[LCSW];
#pragma unroll
  for (int w = 0; w < LCSW; ++w) V[w] = ~0ull;
  for (int t = 0; t < lj; ++t) {
    uint8_t c = sj[t];
    int code = letter_code(c);
    unsigned long long M[LCSW];
    if (code >= 0) {
#pragma unroll
      for (int w = 0; w < LCSW; ++w) M[w] = maskI[code][w];
    } else {  // rare byte (lower case, IUPAC): exact-equality mask on the fly
#pragma unroll
      for (int w = 0; w < LCSW; ++w) M[w] = 0;
      for (int q = 0; q < li; ++q)
        if (si[q] == c) M[q >> 6] |= 1ull << (q & 63);
    }
    unsigned long long carry = 0;
#pragma unroll
    for (int w = 0; w < LCSW; ++w) {
      unsigned long long U = V[w] & M[w];
      unsigned long long s1 = V[w] + U;
      unsigned long long c1 = s1 < V[w];
      unsigned long long s2 = s1 + carry;
      unsigned long long c2 = s2 < s1;
      carry = c1 | c2;
      V[w] = s2 | (V[w] & ~M[w]);
    }
  }
  int zeros = 0;
#pragma unroll
  for (int w = 0; w < LCSW; ++w) {
    int lo = w * 64;
    if (li > lo) {
      int nb = min(64, li - lo);
      unsigned long long keep = (nb == 64) ? ~0ull : ((1ull << nb) - 1ull);
      zeros += __popcll(~V[w] & keep);
    }
  }
  return zeros;
}

// ---- alignment node descriptor ------------------------------------------------
struct Node {
  const uint8_t* p;  // row-major chars
  int rows, len, stride;
};

// ---- profile: src/align.h:131-171, compressed to the non-zero entries ---------
// column record (PROFW dwords): [0] = cnt | k0<<4 | k1<<8 | k2<<12 | k3<<16 | k4<<20, [1..5] = float values
// single-sequence mode: [0] = the raw byte.
__device__ __forceinline__ void build_profile(const Node& a, uint32_t* prof, MsaLds& L, int lane) {
  // first / last aligned nucleotide per row (align.h:139-151)
  for (int i = 0; i < a.rows; ++i) {
    int first = -1, last = a.len;
    for (int base = 0; base < a.len; base += WAVE) {
      int j = base + lane;
      bool nz = (j < a.len) && (a.p[(size_t)i * a.stride + j] != '-');
      unsigned long long bm = __ballot(nz);
      if (bm) {
        if (first == -1) first = base + __builtin_ctzll(bm);
        last = base + 63 - __builtin_clzll(bm);
      }
    }
    if (lane == 0) {
      L.first[i] = first;
      L.last[i] = last;
    }
  }
  __syncthreads();
  for (int j = lane; j < a.len; j += WAVE) {
    float cnt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int sum = 0;
    for (int i = 0; i < a.rows; ++i) {
      int f = L.first[i], l = L.last[i];
      // first == -1 (all-gap row): the reference's test (firstAlignedNuc <= j) is true and
      // lastAlignedNuc stays a.shape()[1], so the row counts everywhere
      if (f <= j && j <= l) {
        ++sum;
        uint8_t ch = a.p[(size_t)i * a.stride + j];
        if (ch == 'A' || ch == 'a') cnt[0] += 1.f;
        else if (ch == 'C' || ch == 'c') cnt[1] += 1.f;
        else if (ch == 'G' || ch == 'g') cnt[2] += 1.f;
        else if (ch == 'T' || ch == 't') cnt[3] += 1.f;
        else if (ch == 'N' || ch == 'n') cnt[4] += 1.f;
        else if (ch == '-') cnt[5] += 1.f;
        else --sum;
      }
    }
    float fs = (float)sum;
    uint32_t meta = 0;
    int n = 0;
    uint32_t* rec = prof + (size_t)j * PROFW;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float p = cnt[k] / fs;  // 0/0 = NaN when only foreign letters cover the column (align.h:169)
      if (!(p == 0.0f)) {     // NaN counts as non-zero
        meta |= (uint32_t)k << (4 + 4 * n);
        rec[1 + n] = __float_as_uint(p);
        ++n;
      }
    }
    rec[0] = meta | (uint32_t)n;
  }
}

__device__ __forceinline__ void build_single(const Node& a, uint32_t* prof, int lane) {
  for (int j = lane; j < a.len; j += WAVE) prof[(size_t)j * PROFW] = a.p[j];
}

// (int) score as x86-64 cvttss2si does it: NaN / out of range -> 0x80000000
__device__ __forceinline__ int cvt_x86(float f) {
  if (!(f == f) || f >= 2147483648.0f || f < -2147483648.0f) return (int)0x80000000;
  return (int)f;
}

// ---- K2: Gotoh DP (gotoh.h:103-141) ------------------------------------------------
// rows = columns of a1 (slot s = row s, slot 0 = border row), columns = columns of a2,
// AlignConfig<true,true> (src/msa.h:106): end gaps free on both sequences.
// Trace nibble per cell: bit0 = bit1, bit1 = bit2, bit2 = bit3, bit3 = bit4 of gotoh.h:88-91.
// Returns S[m][n] (the alignment score) in every lane.
template <int K, bool SINGLE>
__device__ __noinline__ int gotoh_pass(const uint32_t* prof1, const uint32_t* prof2, int m, int n,
                                       const dellyhip_params& P, uint32_t* bits, int lane) {
  int S[K], H[K], hgo[K], hge[K];
  uint32_t rmeta[K];
  float rp[K][5];
  uint32_t accA[K], accB[K];
  int mx1[K];
  const int go = P.gap_open, ge = P.gap_extend;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    S[i] = 0;        // S[r][0] = _verticalGap(ac, 0, n, ...) = 0
    H[i] = -GINF;    // newhoz at column 0
    hgo[i] = (s == m) ? 0 : go + ge;   // _horizontalGap(ac, row, m, .): free in the last row
    hge[i] = (s == m) ? 0 : ge;
    rmeta[i] = SINGLE ? (uint32_t)NOMATCH : 0u;
#pragma unroll
    for (int q = 0; q < 5; ++q) rp[i][q] = 0.f;
    if (s >= 1 && s <= m) {
      const uint32_t* rec = prof1 + (size_t)(s - 1) * PROFW;
      rmeta[i] = rec[0];
      if (!SINGLE) {
#pragma unroll
        for (int q = 0; q < 5; ++q) rp[i][q] = __uint_as_float(rec[1 + q]);
      }
    }
    accA[i] = accB[i] = 0;
    int n1 = SINGLE ? 0 : (int)(rmeta[i] & 15u);
    int mx = 0;
#pragma unroll
    for (int q = 1; q <= 5; ++q)
      if (__ballot(n1 >= q)) mx = q;
    mx1[i] = mx;
  }
  const float fm = (float)P.match, fmm = (float)P.mismatch;
  const int T = n + 63;
  const int nblk = (T + 15) >> 4;
  int upPrevS = 0;
  int vbot = 0;  // V[r][0] = 0 for the lane's last row
  uint32_t cmeta = SINGLE ? (uint32_t)NOMATCH : 0u;
  float cp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int c = -lane;
  for (int blk = 0; blk < nblk; ++blk) {
    int ci = blk * 16 + (lane & 15);
    uint32_t chm = SINGLE ? (uint32_t)NOMATCH : 0u;
    float chp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (ci < n) {
      const uint32_t* rec = prof2 + (size_t)ci * PROFW;
      chm = rec[0];
      if (!SINGLE) {
#pragma unroll
        for (int q = 0; q < 5; ++q) chp[q] = __uint_as_float(rec[1 + q]);
      }
    }
    for (int f = 0; f < 16; ++f) {
      cmeta = (uint32_t)dpp_from_prev((int)cmeta, __builtin_amdgcn_readlane((int)chm, f));
      if (!SINGLE) {
#pragma unroll
        for (int q = 0; q < 5; ++q)
          cp[q] = __int_as_float(
              dpp_from_prev(__float_as_int(cp[q]), __builtin_amdgcn_readlane(__float_as_int(chp[q]), f)));
      }
      const int recvS = dpp_from_prev(S[K - 1], 0);
      const int recvV = dpp_from_prev(vbot, -GINF);
      c += 1;
      const bool active = (unsigned)(c - 1) < (unsigned)n;
      int mx2 = 0;
      if (!SINGLE) {
        int n2a = active ? (int)(cmeta & 15u) : 0;
#pragma unroll
        for (int q = 1; q <= 5; ++q)
          if (__ballot(n2a >= q)) mx2 = q;
      }
      if (active) {
        const int vgo = (c == n) ? 0 : go + ge;   // _verticalGap(ac, col, n, .): free in the last column
        const int vge = (c == n) ? 0 : ge;
        int dS = upPrevS, uS = recvS, uV = recvV;
        const int n2 = (int)(cmeta & 15u);
#pragma unroll
        for (int i = 0; i < K; ++i) {
          int sco;
          if (SINGLE) {
            sco = (rmeta[i] == cmeta) ? P.match : P.mismatch;   // align.h:100-102
          } else {
            float sc = 0.f;
            const int n1 = (int)(rmeta[i] & 15u);
#pragma unroll
            for (int i1 = 0; i1 < 5; ++i1) {
              if (i1 < mx1[i]) {
#pragma unroll
                for (int i2 = 0; i2 < 5; ++i2) {
                  if (i2 < mx2) {
                    const bool on = (i1 < n1) && (i2 < n2);
                    const uint32_t k1 = (rmeta[i] >> (4 + 4 * i1)) & 7u, k2 = (cmeta >> (4 + 4 * i2)) & 7u;
                    const float t = (rp[i][i1] * cp[i2]) * ((k1 == k2) ? fm : fmm);   // align.h:108
                    sc = on ? (sc + t) : sc;
                  }
                }
              }
            }
            sco = cvt_x86(sc);
          }
          const int hext = H[i] + hge[i];
          const int vext = uV + vge;
          const int newhoz = max(S[i] + hgo[i], hext);
          int v = max(uS + vgo, vext);
          int s = max(max((int)((uint32_t)dS + (uint32_t)sco), newhoz), v);
          uint32_t nib = (newhoz != hext ? 1u : 0u) | (v != vext ? 2u : 0u);
          if (s == newhoz) nib |= 4u;
          else if (s == v) nib |= 8u;
          if (i == 0 && lane == 0) {  // slot 0 = border row 0: S = 0 (free end gap), V = -inf
            s = 0;
            v = -GINF;
          }
          dS = S[i];
          uS = s;
          uV = v;
          S[i] = s;
          H[i] = newhoz;
          if (f < 8) accA[i] |= nib << (4 * f);
          else accB[i] |= nib << (4 * (f - 8));
        }
        vbot = uV;
      }
      upPrevS = recvS;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      bits[((size_t)(blk * 2 + 0) * K + i) * WAVE + lane] = accA[i];
      bits[((size_t)(blk * 2 + 1) * K + i) * WAVE + lane] = accB[i];
      accA[i] = accB[i] = 0;
    }
  }
  int fin = 0;
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (lane * K + i == m) fin = S[i];
  return __shfl(fin, m / K);
}

// traceback state machine of gotoh.h:143-167 over the stored nibbles (uniform, serial)
template <int K>
__device__ __noinline__ int gotoh_traceback(const uint32_t* bits, int row, int col, uint8_t* tr, int lane, int& tailV,
                                            int& tailH) {
  int tl = 0;
  int state = 0;  // 0 's', 1 'h', 2 'v'
  while (row > 0 && col > 0) {
    int l = row / K, i = row - l * K;
    int t0 = col + l - 1;
    uint32_t w = ld_scratch(&bits[((size_t)(t0 >> 3) * K + i) * WAVE + l]);
    w = (uint32_t)rfl((int)w);
    uint32_t nib = (w >> (4 * (t0 & 7))) & 15u;
    if (state == 0) {
      if (nib & 4u) state = 1;
      else if (nib & 8u) state = 2;
      else {
        --row;
        --col;
        if (lane == 0) tr[tl] = 0;
        ++tl;
        continue;
      }
    }
    if (state == 1) {
      if (nib & 1u) state = 0;
      --col;
      if (lane == 0) tr[tl] = 2;
      ++tl;
    } else {
      if (nib & 2u) state = 0;
      --row;
      if (lane == 0) tr[tl] = 1;
      ++tl;
    }
  }
  tailV = (col == 0) ? row : 0;
  tailH = (row == 0) ? col : 0;
  return tl;
}

template <int K>
__device__ __forceinline__ int gotoh_dispatch_k(bool single, const uint32_t* p1, const uint32_t* p2, int m, int n,
                                                const dellyhip_params& P, uint32_t* bits, uint8_t* tr, int lane,
                                                int& tl, int& tailV, int& tailH) {
  int score = single ? gotoh_pass<K, true>(p1, p2, m, n, P, bits, lane) : gotoh_pass<K, false>(p1, p2, m, n, P, bits, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tl = gotoh_traceback<K>(bits, m, n, tr, lane, tailV, tailH);
  return score;
}

// gotoh(a1, a2, align, AlignConfig<true,true>, sc): merges two nodes into `out`
// (rows a1 then rows a2, row stride LCAP).  Returns 0 or DELLYHIP_E_LIMIT.
__device__ __forceinline__ int merge_nodes(const Node& a1, const Node& a2, uint8_t* out, int& out_len, int& score,
                                           const dellyhip_params& P, uint32_t* prof, uint32_t* bits, MsaLds& L,
                                           int lane) {
  const int m = a1.len, n = a2.len;
  if (m > LCAP - 1 || n > LCAP || m + 1 > WAVE * GKMAX) return DELLYHIP_E_LIMIT;
  uint32_t* p1 = prof;
  uint32_t* p2 = prof + (size_t)LCAP * PROFW;
  const bool single = (a1.rows == 1 && a2.rows == 1);
  if (single) {
    build_single(a1, p1, lane);
    build_single(a2, p2, lane);
  } else {
    build_profile(a1, p1, L, lane);
    __syncthreads();
    build_profile(a2, p2, L, lane);
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int tl = 0, tailV = 0, tailH = 0;
  const int K = (m + 1 + WAVE - 1) / WAVE;
  switch (K) {
    case 1: score = gotoh_dispatch_k<1>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 2: score = gotoh_dispatch_k<2>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 3: score = gotoh_dispatch_k<3>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 4: score = gotoh_dispatch_k<4>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 5: score = gotoh_dispatch_k<5>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 6: score = gotoh_dispatch_k<6>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    case 7: score = gotoh_dispatch_k<7>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
    default: score = gotoh_dispatch_k<8>(single, p1, p2, m, n, P, bits, L.trace, lane, tl, tailV, tailH); break;
  }
  __syncthreads();
  const int tail = tailV + tailH;
  const int alen = tail + tl;
  out_len = alen;
  if (alen > LCAP) return DELLYHIP_E_LIMIT;
  // _createAlignment align.h:202-229: columns = reversed trace (tail first)
  int c1 = tailV, c2 = tailH;  // bases of a1 / a2 consumed before the recorded part
  for (int j = lane; j < tail; j += WAVE) {
    for (int i = 0; i < a1.rows; ++i) out[(size_t)i * LCAP + j] = tailV ? a1.p[(size_t)i * a1.stride + j] : '-';
    for (int i = 0; i < a2.rows; ++i) out[(size_t)(a1.rows + i) * LCAP + j] = tailH ? a2.p[(size_t)i * a2.stride + j] : '-';
  }
  for (int base = 0; base < tl; base += WAVE) {
    int q = base + lane;
    int op = (q < tl) ? (int)L.trace[tl - 1 - q] : 0;
    unsigned long long mv = __ballot(q < tl && op != 2);  // consumes a1 column
    unsigned long long mr = __ballot(q < tl && op != 1);  // consumes a2 column
    unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int i1 = c1 + __popcll(mv & below), i2 = c2 + __popcll(mr & below);
    if (q < tl) {
      int j = tail + q;
      for (int i = 0; i < a1.rows; ++i) out[(size_t)i * LCAP + j] = (op != 2) ? a1.p[(size_t)i * a1.stride + i1] : '-';
      for (int i = 0; i < a2.rows; ++i)
        out[(size_t)(a1.rows + i) * LCAP + j] = (op != 1) ? a2.p[(size_t)i * a2.stride + i2] : '-';
    }
    c1 += __popcll(mv);
    c2 += __popcll(mr);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return 0;
}

// ---- K7: consensus  src/msa.h:111-173.  Writes the ungapped consensus (<= cap bytes), returns its length.
__device__ __forceinline__ int consensus_node(const Node& a, const dellyhip_params& P, uint8_t* cs, int cap, MsaLds& L,
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
  __syncthreads();
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

// ---- msa() for one junction ----------------------------------------------------
__device__ void msa_junction(const MsaArgs& A, int j, MsaLds& L, uint8_t* ws, int nmax, int lane) {
  const dellyhip_junction J = A.junc[j];
  dellyhip_result* out = &A.res[j];
  uint8_t* cons_out = A.out_blob + (size_t)j * A.out_stride;
  const int N = J.n_seq;
  int status = 0, cons_len = 0, rows = 0;
  // shortpe.h:166-171: junctions with <= 1 read get no consensus
  if (N >= 2) {
    if (N > NRMAX || N > nmax) status = DELLYHIP_E_LIMIT;
    const uint64_t o0 = A.seq_off[J.seq_first];
    if (!status) {
      for (int r = lane; r < N; r += WAVE) {
        uint64_t a = A.seq_off[J.seq_first + r], b = A.seq_off[J.seq_first + r + 1];
        L.roff[r] = (uint32_t)(a - o0);
        L.rlen[r] = (int32_t)(b - a);
      }
      __syncthreads();
      int bad = 0;
      for (int r = 0; r < N; ++r)
        if (L.rlen[r] > RLMAX || L.rlen[r] < 1) bad = 1;
      if (bad) status = DELLYHIP_E_LIMIT;
    }
    if (!status) {
      const uint8_t* blob = A.seq_blob + o0;
      // --- distanceMatrix (msa.h:32-44): match masks, then one pair per lane
      for (int q = lane; q < N * 5 * LCSW; q += WAVE) (&L.lcsmask[0][0][0])[q] = 0ull;
      __syncthreads();
      for (int r = 0; r < N; ++r) {
        const uint8_t* s = blob + L.roff[r];
        for (int base = 0; base < L.rlen[r]; base += WAVE) {
          int q = base + lane;
          int code = (q < L.rlen[r]) ? letter_code(s[q]) : -1;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            unsigned long long bm = __ballot(code == k);
            if (lane == 0) L.lcsmask[r][k][base >> 6] = bm;
          }
        }
      }
      const int D = NODES;
      for (int q = lane; q < D * D; q += WAVE) {
        int i = q / D, jj = q - i * D;
        L.d[q] = (jj > i) ? (int8_t)-1 : (int8_t)0;
      }
      for (int q = lane; q < D; q += WAVE) {
        L.par[q] = -1;
        L.lch[q] = -1;
        L.rch[q] = -1;
      }
      __syncthreads();
      const int npairs = N * (N - 1) / 2;
      for (int pbase = 0; pbase < npairs; pbase += WAVE) {
        int pi = pbase + lane;
        if (pi < npairs) {
          // pair index -> (i, jj), i < jj
          int i = 0, rem = pi;
          while (rem >= N - 1 - i) {
            rem -= N - 1 - i;
            ++i;
          }
          int jj = i + 1 + rem;
          int l = lcs_bitparallel(L.lcsmask[i], blob + L.roff[i], L.rlen[i], blob + L.roff[jj], L.rlen[jj]);
          int mn = min(L.rlen[i], L.rlen[jj]);
          L.d[i * D + jj] = (int8_t)((l * 100) / mn);   // msa.h:41
        }
      }
      __syncthreads();
      // --- upgma (msa.h:46-89)
      int nn = N;
      for (; nn < 2 * N + 1; ++nn) {
        int key = -1;
        for (int q = lane; q < nn * D; q += WAVE) {   // rows 0..nn-1
          int i = q / D, jj = q - i * D;
          if (jj > i && jj < nn) {
            int dv = L.d[q];
            if (dv > -1) {
              int k2 = ((dv + 1) << 13) | (8191 - q);   // max d, then first in row-major order
              key = max(key, k2);
            }
          }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o));
        if (key < 0) break;
        int q = 8191 - (key & 8191);
        int dI = q / D, dJ = q - dI * D;
        __syncthreads();
        if (lane == 0) {
          L.par[dI] = (int16_t)nn;
          L.par[dJ] = (int16_t)nn;
          L.lch[nn] = (int16_t)dI;
          L.rch[nn] = (int16_t)dJ;
        }
        __syncthreads();
        for (int i = lane; i < nn; i += WAVE) {
          if (L.par[i] == -1) {
            int a = (dI < i) ? L.d[dI * D + i] : L.d[i * D + dI];
            int b = (dJ < i) ? L.d[dJ * D + i] : L.d[i * D + dJ];
            L.d[i * D + nn] = (int8_t)((a + b) / 2);
          }
        }
        __syncthreads();
        for (int i = lane; i < nn + 1; i += WAVE) {
          if (i < dI) L.d[i * D + dI] = -1;
          if (i > dI) L.d[dI * D + i] = -1;
          if (i < dJ) L.d[i * D + dJ] = -1;
          if (i > dJ) L.d[dJ * D + i] = -1;
        }
        __syncthreads();
      }
      const int root = (nn > 0) ? nn - 1 : 0;
      // --- palign (msa.h:91-109): internal nodes in creation order (children first)
      uint8_t* alnbuf = ws;
      uint32_t* prof = reinterpret_cast<uint32_t*>(ws + ((MsaWs::node_rows_cap(nmax) * LCAP + 255) & ~255ull));
      uint32_t* bits = prof + 2 * (size_t)LCAP * PROFW;
      if (lane == 0) {
        int base = 0;
        for (int r = 0; r < N; ++r) {
          L.node_rows[r] = 1;
          L.node_len[r] = L.rlen[r];
          L.node_base[r] = -1;
        }
        for (int x = N; x <= root; ++x) {
          L.node_rows[x] = L.node_rows[L.lch[x]] + L.node_rows[L.rch[x]];
          L.node_base[x] = base;
          base += L.node_rows[x];
        }
      }
      __syncthreads();
      for (int x = N; x <= root && !status; ++x) {
        int lc = L.lch[x], rc = L.rch[x];
        Node a1, a2;
        a1.rows = L.node_rows[lc]; a1.len = L.node_len[lc];
        a2.rows = L.node_rows[rc]; a2.len = L.node_len[rc];
        if (lc < N) { a1.p = blob + L.roff[lc]; a1.stride = 0; }
        else { a1.p = alnbuf + (size_t)L.node_base[lc] * LCAP; a1.stride = LCAP; }
        if (rc < N) { a2.p = blob + L.roff[rc]; a2.stride = 0; }
        else { a2.p = alnbuf + (size_t)L.node_base[rc] * LCAP; a2.stride = LCAP; }
        int olen = 0, score = 0;
        int rcode = merge_nodes(a1, a2, alnbuf + (size_t)L.node_base[x] * LCAP, olen, score, A.p, prof, bits, L, lane);
        if (rcode) status = rcode;
        if (lane == 0) L.node_len[x] = olen;
        __syncthreads();
      }
      if (!status) {
        Node r;
        r.rows = L.node_rows[root];
        r.len = L.node_len[root];
        if (root < N) { r.p = blob + L.roff[root]; r.stride = 0; }
        else { r.p = alnbuf + (size_t)L.node_base[root] * LCAP; r.stride = LCAP; }
        cons_len = consensus_node(r, A.p, cons_out, OUT_CONS_CAP, L, lane);
        rows = r.rows;
      }
    }
  }
  if (lane == 0) {
    out->sr_support = rows;
    out->status = status;
    A.cons_len[j] = status ? 0 : cons_len;
  }
  __syncthreads();
}

__global__ __launch_bounds__(WAVE) void msa_kernel(MsaArgs A, int nmax) {
  __shared__ MsaLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    msa_junction(A, w, L, ws, nmax, lane);
  }
}

// single gotoh(a1, a2) on caller-supplied alignments (dellyhip_gotoh)
__global__ __launch_bounds__(WAVE) void gotoh_single_kernel(MsaArgs A) {
  __shared__ MsaLds L;
  const int lane = threadIdx.x;
  Node a1{A.g_a1, A.g_r1, A.g_m, A.g_m}, a2{A.g_a2, A.g_r2, A.g_n, A.g_n};
  uint32_t* prof = reinterpret_cast<uint32_t*>(A.ws);
  uint32_t* bits = prof + 2 * (size_t)LCAP * PROFW;
  int olen = 0, score = 0;
  int rc = (A.g_r1 + A.g_r2 > 2 * NRMAX || A.g_r1 > NRMAX || A.g_r2 > NRMAX) ? DELLYHIP_E_LIMIT
                                                                               : merge_nodes(a1, a2, A.g_out, olen, score, A.p, prof, bits, L, lane);
  if (lane == 0) {
    A.g_info[0] = olen;
    A.g_info[1] = score;
    A.g_info[2] = rc;
  }
}

__global__ void lcs_single_kernel(const uint8_t* s1, int m, const uint8_t* s2, int n, int* out) {
  __shared__ unsigned long long mask[5][LCSW];
  const int lane = threadIdx.x;
  for (int base = 0; base < LCSW * 64; base += WAVE) {
    int q = base + lane;
    int code = (q < m) ? letter_code(s1[q]) : -1;
    for (int k = 0; k < 5; ++k) {
      unsigned long long bm = __ballot(code == k);
      if (lane == 0) mask[k][base >> 6] = bm;
    }
  }
  __syncthreads();
  if (lane == 0) *out = lcs_bitparallel(mask, s1, m, s2, n);
}

// ---- host helpers -----------------------------------------------------------------
inline int msa_prepare(const std::vector<dellyhip_junction>& junc, const uint64_t* seq_off, uint64_t& ws_stride,
                       int* nmax_out = nullptr) {
  int nmax = 2;
  for (auto const& J : junc) nmax = std::max(nmax, std::min<int>(J.n_seq, NRMAX));
  (void)seq_off;
  ws_stride = (MsaWs::bytes(nmax) + 255) & ~255ull;
  if (nmax_out) *nmax_out = nmax;
  return 0;
}

inline int msa_launch(const MsaArgs& a, int grid, int nmax, hipStream_t s) {
  hipLaunchKernelGGL(msa_kernel, dim3(grid), dim3(WAVE), 0, s, a, nmax);
  return 0;
}

inline int msa_single_lcs(hipStream_t s, const char* s1, int m, const char* s2, int n, int32_t* out) {
  if (m > RLMAX || m < 0 || n < 0) return DELLYHIP_E_LIMIT;
  uint8_t *d1 = nullptr, *d2 = nullptr;
  int* dout = nullptr;
  if (hipMalloc((void**)&d1, std::max(m, 1)) != hipSuccess || hipMalloc((void**)&d2, std::max(n, 1)) != hipSuccess ||
      hipMalloc((void**)&dout, 4) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(d1, s1, m, hipMemcpyHostToDevice);
  (void)hipMemcpy(d2, s2, n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(lcs_single_kernel, dim3(1), dim3(WAVE), 0, s, d1, m, d2, n, dout);
  hipError_t e = hipStreamSynchronize(s);
  (void)hipMemcpy(out, dout, 4, hipMemcpyDeviceToHost);
  (void)hipFree(d1);
  (void)hipFree(d2);
  (void)hipFree(dout);
  return e == hipSuccess ? 0 : DELLYHIP_E_RUNTIME;
}

inline int msa_single_gotoh(hipStream_t s, const dellyhip_params& P, const char* a1, int r1, int m, const char* a2,
                            int r2, int n, char* out, int cap, int32_t* len, int32_t* score) {
  if (r1 < 1 || r2 < 1 || r1 > NRMAX || r2 > NRMAX || m < 0 || n < 0 || m > LCAP - 1 || n > LCAP) return DELLYHIP_E_LIMIT;
  uint8_t *d1 = nullptr, *d2 = nullptr, *dout = nullptr, *ws = nullptr;
  int* dinfo = nullptr;
  size_t wsb = MsaWs::bytes(2);
  if (hipMalloc((void**)&d1, std::max(r1 * m, 1)) != hipSuccess || hipMalloc((void**)&d2, std::max(r2 * n, 1)) != hipSuccess ||
      hipMalloc((void**)&dout, (size_t)(r1 + r2) * LCAP) != hipSuccess || hipMalloc((void**)&ws, wsb) != hipSuccess ||
      hipMalloc((void**)&dinfo, 16) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(d1, a1, (size_t)r1 * m, hipMemcpyHostToDevice);
  (void)hipMemcpy(d2, a2, (size_t)r2 * n, hipMemcpyHostToDevice);
  MsaArgs A{};
  A.p = P;
  A.ws = ws;
  A.g_a1 = d1; A.g_a2 = d2; A.g_r1 = r1; A.g_m = m; A.g_r2 = r2; A.g_n = n;
  A.g_out = dout;
  A.g_info = dinfo;
  hipLaunchKernelGGL(gotoh_single_kernel, dim3(1), dim3(WAVE), 0, s, A);
  hipError_t e = hipStreamSynchronize(s);
  int info[3] = {0, 0, DELLYHIP_E_RUNTIME};
  if (e == hipSuccess) (void)hipMemcpy(info, dinfo, 12, hipMemcpyDeviceToHost);
  int rc = info[2];
  if (!rc) {
    *len = info[0];
    *score = info[1];
    if (info[0] > cap) rc = DELLYHIP_E_ARG;
    else {
      std::vector<uint8_t> tmp((size_t)(r1 + r2) * LCAP);
      (void)hipMemcpy(tmp.data(), dout, tmp.size(), hipMemcpyDeviceToHost);
      for (int i = 0; i < r1 + r2; ++i) memcpy(out + (size_t)i * cap, tmp.data() + (size_t)i * LCAP, info[0]);
    }
  }
  (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dout); (void)hipFree(ws); (void)hipFree(dinfo);
  return rc;
}

// msa(c, sps, cs) for one read set (dellyhip_msa)
inline int msa_single(hipStream_t s, const dellyhip_params& P, int n_reads, const char* seq_blob,
                      const uint64_t* seq_off, char* cs, int cs_cap, int32_t* cs_len, int32_t* rows) {
  if (n_reads > NRMAX) return DELLYHIP_E_LIMIT;
  dellyhip_junction J{};
  J.n_seq = n_reads;
  J.seq_first = 0;
  uint64_t blob_bytes = n_reads ? seq_off[n_reads] : 0;
  int nmax = std::max(2, n_reads);
  uint64_t wsb = (MsaWs::bytes(nmax) + 255) & ~255ull;
  dellyhip_junction* dj = nullptr;
  uint8_t *dblob = nullptr, *dout = nullptr, *ws = nullptr;
  uint64_t* doff = nullptr;
  dellyhip_result* dres = nullptr;
  int32_t *dlen = nullptr, *dcnt = nullptr;
  if (hipMalloc((void**)&dj, sizeof J) != hipSuccess || hipMalloc((void**)&dblob, std::max<uint64_t>(blob_bytes, 1)) != hipSuccess ||
      hipMalloc((void**)&doff, (n_reads + 1) * 8) != hipSuccess || hipMalloc((void**)&dres, sizeof(dellyhip_result)) != hipSuccess ||
      hipMalloc((void**)&dout, LCAP) != hipSuccess || hipMalloc((void**)&ws, wsb) != hipSuccess ||
      hipMalloc((void**)&dlen, 4) != hipSuccess || hipMalloc((void**)&dcnt, 4) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(dj, &J, sizeof J, hipMemcpyHostToDevice);
  (void)hipMemcpy(dblob, seq_blob, blob_bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(doff, seq_off, (n_reads + 1) * 8, hipMemcpyHostToDevice);
  (void)hipMemset(dres, 0, sizeof(dellyhip_result));
  (void)hipMemset(dcnt, 0, 4);
  (void)hipMemset(dlen, 0, 4);
  MsaArgs A{};
  A.junc = dj; A.seq_blob = dblob; A.seq_off = doff; A.p = P; A.res = dres; A.out_blob = dout; A.out_stride = LCAP;
  A.cons_len = dlen; A.ws = ws; A.ws_stride = wsb; A.n_work = 1; A.work_counter = dcnt;
  hipLaunchKernelGGL(msa_kernel, dim3(1), dim3(WAVE), 0, s, A, nmax);
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
      if (L > cs_cap || L > OUT_CONS_CAP) rc = (L > OUT_CONS_CAP) ? DELLYHIP_E_LIMIT : DELLYHIP_E_ARG;
      else (void)hipMemcpy(cs, dout, L, hipMemcpyDeviceToHost);
    }
  }
  (void)hipFree(dj); (void)hipFree(dblob); (void)hipFree(doff); (void)hipFree(dres); (void)hipFree(dout);
  (void)hipFree(ws); (void)hipFree(dlen); (void)hipFree(dcnt);
  return rc;
}

}  // namespace dh
