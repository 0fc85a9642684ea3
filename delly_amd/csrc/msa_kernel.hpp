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
constexpr int TMAXC = 96;    // profile column types per alignment node the score table is built for
constexpr int HSLOTS = 256;  // open-addressing table of column-type keys
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

// guide-tree phase (distanceMatrix + upgma) and alignment phase never overlap
struct MsaLdsTree {
  unsigned long long lcsmask[NRMAX][5][LCSW];
  int8_t d[NODES * NODES];
};
struct MsaLdsLut {
  unsigned long long hkey[HSLOTS];          // open-addressing set of column-type keys
  unsigned long long tkey[2][TMAXC];        // key of each dense type id, per profile
  uint8_t slot_id[HSLOTS];
  uint8_t type[2][LCAP];                    // column -> type id
  int8_t tab[TMAXC * TMAXC];                // (int) score of (type1, type2)
};
struct __attribute__((aligned(16))) MsaLds {
  union {
    MsaLdsTree t;
    MsaLdsLut g;
  } u;
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
// first / last aligned nucleotide per row (align.h:139-151) -> L.first / L.last
__device__ __forceinline__ void row_spans(const Node& a, MsaLds& L, int lane) {
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
}

// letter counts of column j over the rows that cover it (align.h:153-166); cnt[5] = '-'
__device__ __forceinline__ int column_counts(const Node& a, const MsaLds& L, int j, int (&cnt)[6]) {
  int sum = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) cnt[k] = 0;
  for (int i = 0; i < a.rows; ++i) {
    int f = L.first[i], l = L.last[i];
    // first == -1 (all-gap row): the reference's test (firstAlignedNuc <= j) is true and
    // lastAlignedNuc stays a.shape()[1], so the row counts everywhere
    if (f <= j && j <= l) {
      ++sum;
      uint8_t ch = a.p[(size_t)i * a.stride + j];
      if (ch == 'A' || ch == 'a') ++cnt[0];
      else if (ch == 'C' || ch == 'c') ++cnt[1];
      else if (ch == 'G' || ch == 'g') ++cnt[2];
      else if (ch == 'T' || ch == 't') ++cnt[3];
      else if (ch == 'N' || ch == 'n') ++cnt[4];
      else if (ch == '-') ++cnt[5];
      else --sum;
    }
  }
  return sum;
}

__device__ __forceinline__ void build_profile(const Node& a, uint32_t* prof, MsaLds& L, int lane) {
  row_spans(a, L, lane);
  for (int j = lane; j < a.len; j += WAVE) {
    int ic[6];
    const int sum = column_counts(a, L, j, ic);
    float cnt[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cnt[k] = (float)ic[k];
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

// ---- profile column TYPES and their score table --------------------------------------
// A profile column is the float vector count[k]/sum (k = A,C,G,T,N); _score (align.h:104-110)
// of two columns depends only on the two (count[0..4], sum) tuples.  A node of r <= 32 rows has
// few distinct tuples (coverage level x letter, plus the odd mismatch column), so the (int)
// score of every type pair is evaluated once per merge -- with the reference's float
// expression and evaluation order -- into an int8 table in LDS and the DP cell reads one byte.
// key = count[0..4] (6 bits each) | sum << 30
__device__ __forceinline__ unsigned long long column_key(const int (&cnt)[6], int sum) {
  return (unsigned long long)cnt[0] | ((unsigned long long)cnt[1] << 6) | ((unsigned long long)cnt[2] << 12) |
         ((unsigned long long)cnt[3] << 18) | ((unsigned long long)cnt[4] << 24) | ((unsigned long long)sum << 30);
}

// types of node `a` into L.u.g.type[which] / tkey[which]; returns the number of types or -1
// when there are more than tmax
__device__ __forceinline__ int build_types(const Node& a, int which, MsaLds& L, int lane, int tmax) {
  MsaLdsLut& G = L.u.g;
  row_spans(a, L, lane);
  for (int q = lane; q < HSLOTS; q += WAVE) G.hkey[q] = ~0ull;
  __syncthreads();
  int fail = 0;
  for (int j = lane; j < a.len; j += WAVE) {
    int ic[6];
    const int sum = column_counts(a, L, j, ic);
    const unsigned long long key = column_key(ic, sum);
    int slot = (int)((key * 0x9E3779B97F4A7C15ull) >> 56);
    int probes = 0;
    for (; probes < HSLOTS; ++probes) {
      const unsigned long long old = atomicCAS(&G.hkey[slot], ~0ull, key);
      if (old == ~0ull || old == key) break;
      slot = (slot + 1) & (HSLOTS - 1);
    }
    if (probes >= HSLOTS) fail = 1;
    G.type[which][j] = (uint8_t)slot;
  }
  __syncthreads();
  int T = 0;
#pragma unroll
  for (int q = 0; q < HSLOTS / WAVE; ++q) {
    const int sl = q * WAVE + lane;
    const unsigned long long k = G.hkey[sl];
    const bool occ = k != ~0ull;
    const unsigned long long b = __ballot(occ);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int id = T + __popcll(b & below);
    if (occ) {
      G.slot_id[sl] = (uint8_t)min(id, 255);
      if (id < TMAXC) G.tkey[which][id] = k;
    }
    T += __popcll(b);
  }
  if (__ballot(fail) != 0ull) T = HSLOTS + 1;
  __syncthreads();
  if (T > tmax || T > TMAXC) return -1;
  for (int j = lane; j < a.len; j += WAVE) G.type[which][j] = G.slot_id[G.type[which][j]];
  __syncthreads();
  return T;
}

// (int) score as x86-64 cvttss2si does it: NaN / out of range -> 0x80000000
__device__ __forceinline__ int cvt_x86(float f) {
  if (!(f == f) || f >= 2147483648.0f || f < -2147483648.0f) return (int)0x80000000;
  return (int)f;
}

// tab[t1*T2 + t2] = (int) _score(column of type t1, column of type t2)   align.h:104-110
// Returns false when a score is 0x80000000 (NaN profile entry: a column covered by foreign letters
// only) -- not representable in the int8 table, the junction goes to the direct-float kernel.
__device__ __forceinline__ bool fill_table(int T1, int T2, const dellyhip_params& P, MsaLds& L, int lane) {
  MsaLdsLut& G = L.u.g;
  const float fm = (float)P.match, fmm = (float)P.mismatch;
  int nan = 0;
  for (int e = lane; e < T1 * T2; e += WAVE) {
    const int t1 = e / T2, t2 = e - t1 * T2;
    const unsigned long long k1 = G.tkey[0][t1], k2 = G.tkey[1][t2];
    const float s1 = (float)(int)(k1 >> 30), s2 = (float)(int)(k2 >> 30);
    float sc = 0.f;
    for (int a = 0; a < 5; ++a) {
      const float p1 = (float)(int)((k1 >> (6 * a)) & 63ull) / s1;   // align.h:169 (0/0 = NaN)
      if (p1 == 0.0f) continue;                                      // exact: x + (+-0) == x
      for (int b = 0; b < 5; ++b) {
        const float p2 = (float)(int)((k2 >> (6 * b)) & 63ull) / s2;
        if (p2 == 0.0f) continue;
        sc = sc + (p1 * p2) * ((a == b) ? fm : fmm);                 // align.h:108
      }
    }
    const int v = cvt_x86(sc);
    if (v == (int)0x80000000) nan = 1;
    G.tab[e] = (int8_t)v;
  }
  __syncthreads();
  return __ballot(nan) == 0ull;
}

// ---- K2: Gotoh DP (gotoh.h:103-141) ------------------------------------------------
// rows = columns of a1 (slot s = row s, slot 0 = border row), columns = columns of a2,
// AlignConfig<true,true> (src/msa.h:106): end gaps free on both sequences.
// Trace nibble per cell: bit0 = bit1, bit1 = bit2, bit2 = bit3, bit3 = bit4 of gotoh.h:88-91.
// Returns S[m][n] (the alignment score) in every lane.
// MODE 0: both nodes are single sequences (byte compare, align.h:100-102)
// MODE 1: profile x profile through the type-pair score table in LDS (L.u.g)
// MODE 2: profile x profile, float expression evaluated per cell (any number of column types)
template <int K, int MODE>
__device__ __forceinline__ int gotoh_pass_impl(const uint32_t* prof1, const uint32_t* prof2, int m, int n,
                                               const dellyhip_params& P, uint32_t* bits, const MsaLds& L, int T2,
                                               int lane) {
  constexpr bool SINGLE = (MODE == 0);
  constexpr bool FLT = (MODE == 2);
  int S[K], H[K];
  uint32_t rmeta[K];
  float rp[FLT ? K : 1][5];
  uint32_t accA[K], accB[K];
  int mx1[FLT ? K : 1];
  const int go = P.gap_open, ge = P.gap_extend;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    S[i] = 0;        // S[r][0] = _verticalGap(ac, 0, n, ...) = 0
    H[i] = -GINF;    // newhoz at column 0
    rmeta[i] = SINGLE ? (uint32_t)NOMATCH : 0u;
    if (FLT) {
#pragma unroll
      for (int q = 0; q < 5; ++q) rp[i][q] = 0.f;
    }
    if (s >= 1 && s <= m) {
      if (MODE == 1) rmeta[i] = (uint32_t)L.u.g.type[0][s - 1] * (uint32_t)T2;
      else {
        const uint32_t* rec = prof1 + (size_t)(s - 1) * PROFW;
        rmeta[i] = rec[0];
        if (FLT) {
#pragma unroll
          for (int q = 0; q < 5; ++q) rp[i][q] = __uint_as_float(rec[1 + q]);
        }
      }
    }
    accA[i] = accB[i] = 0;
    if (FLT) {
      int n1 = (int)(rmeta[i] & 15u);
      int mx = 0;
#pragma unroll
      for (int q = 1; q <= 5; ++q)
        if (__ballot(n1 >= q)) mx = q;
      mx1[i] = mx;
    }
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
      if (MODE == 1) chm = (uint32_t)L.u.g.type[1][ci];
      else {
        const uint32_t* rec = prof2 + (size_t)ci * PROFW;
        chm = rec[0];
        if (FLT) {
#pragma unroll
          for (int q = 0; q < 5; ++q) chp[q] = __uint_as_float(rec[1 + q]);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      cmeta = (uint32_t)dpp_from_prev((int)cmeta, __builtin_amdgcn_readlane((int)chm, f));
      if (FLT) {
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
      if (FLT) {
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
          } else if (MODE == 1) {
            sco = (int)L.u.g.tab[rmeta[i] + cmeta];   // (NaN scores never reach this kernel: fill_table)
          } else {
            float sc = 0.f;
            const int n1 = (int)(rmeta[i] & 15u);
#pragma unroll
            for (int i1 = 0; i1 < 5; ++i1) {
              if (i1 < mx1[FLT ? i : 0]) {
#pragma unroll
                for (int i2 = 0; i2 < 5; ++i2) {
                  if (i2 < mx2) {
                    const bool on = (i1 < n1) && (i2 < n2);
                    const uint32_t k1 = (rmeta[i] >> (4 + 4 * i1)) & 7u, k2 = (cmeta >> (4 + 4 * i2)) & 7u;
                    const float t = (rp[FLT ? i : 0][i1] * cp[i2]) * ((k1 == k2) ? fm : fmm);   // align.h:108
                    sc = on ? (sc + t) : sc;
                  }
                }
              }
            }
            sco = cvt_x86(sc);
          }
          const bool lastrow = (lane * K + i == m);   // _horizontalGap(ac, row, m, .): free in the last row
          const int hgo = lastrow ? 0 : go + ge, hge = lastrow ? 0 : ge;
          const int hext = H[i] + hge;
          const int vext = uV + vge;
          const int newhoz = max(S[i] + hgo, hext);
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

// out-of-line instance (direct-float kernel, single-item wrapper); the score-table kernel
// inlines the pass so that its __launch_bounds__ register budget covers it
template <int K, int MODE>
__device__ __noinline__ int gotoh_pass(const uint32_t* prof1, const uint32_t* prof2, int m, int n,
                                       const dellyhip_params& P, uint32_t* bits, const MsaLds& L, int T2, int lane) {
  return gotoh_pass_impl<K, MODE>(prof1, prof2, m, n, P, bits, L, T2, lane);
}

// traceback state machine of gotoh.h:143-167 over the stored nibbles (uniform).  Nibble
// words are fetched in windows -- lane l loads the word of cell (row-l, col-l) -- and the walk
// runs out of registers while the path stays inside the fetched words (8 columns per row).
// In state 's' a run of diagonal cells (neither bit3 nor bit4 set) is taken in one step: each
// lane decodes the cell of its own row on the current diagonal, a ballot gives the run length.
template <int K>
__device__ __noinline__ int gotoh_traceback(const uint32_t* bits, int row, int col, uint8_t* tr, int lane, int& tailV,
                                            int& tailH) {
  int tl = 0;
  int state = 0;  // 0 's', 1 'h', 2 'v'
  row = rfl(row);
  col = rfl(col);
  while (row > 0 && col > 0) {
    const int r = row - lane, c = col - lane;
    const int lo = (r >= 1) ? r / K : 0, ii = r - lo * K;
    uint32_t w = 0;
    int tw = -1;
    if (r >= 1 && c >= 1) {
      tw = (c + lo - 1) >> 3;
      w = ld_scratch(&bits[((size_t)tw * K + ii) * WAVE + lo]);
    }
    int l = 0;
    bool inwin = true;
    while (inwin) {
      const int d = lane - l;
      const int cx = col - d;
      const int tx = cx + lo - 1;
      const bool valid = (d >= 0) && (r >= 1) && (cx >= 1) && ((tx >> 3) == tw);
      const uint32_t nib = valid ? ((w >> (4 * (tx & 7))) & 15u) : 16u;
      if (state == 0) {
        const unsigned long long dm = __ballot(valid && (nib & 12u) == 0u) >> l;
        const int L = (~dm == 0ull) ? WAVE : __builtin_ctzll(~dm);
        if (L > 0) {
          if (d >= 0 && d < L) tr[tl + d] = 0;
          tl += L;
          row -= L;
          col -= L;
          l += L;
        }
      }
      if (row <= 0 || col <= 0 || l >= WAVE) {
        inwin = false;
      } else {
        const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)nib, l);
        if (nl == 16u) {
          inwin = false;   // outside the fetched word of this row
        } else {
          if (state == 0) state = (nl & 4u) ? 1 : 2;   // (a diagonal cell cannot reach this point)
          if (state == 1) {
            if (nl & 1u) state = 0;
            --col;
            if (lane == 0) tr[tl] = 2;
            ++tl;
          } else {
            if (nl & 2u) state = 0;
            --row;
            ++l;
            if (lane == 0) tr[tl] = 1;
            ++tl;
          }
          if (row <= 0 || col <= 0 || l >= WAVE) inwin = false;
        }
      }
    }
  }
  tailV = (col == 0) ? row : 0;
  tailH = (row == 0) ? col : 0;
  return tl;
}

// mode: 0 single x single, 1 score table, 2 direct float
template <int K, bool SLOW>
__device__ __forceinline__ int gotoh_dispatch_k(int mode, const uint32_t* p1, const uint32_t* p2, int m, int n,
                                                const dellyhip_params& P, uint32_t* bits, MsaLds& L, int T2, int lane,
                                                int& tl, int& tailV, int& tailH) {
  int score;
  if constexpr (SLOW) {
    if (mode == 0) score = gotoh_pass<K, 0>(p1, p2, m, n, P, bits, L, T2, lane);
    else score = gotoh_pass<K, 2>(p1, p2, m, n, P, bits, L, T2, lane);
  } else {
    if (mode == 0) score = gotoh_pass_impl<K, 0>(p1, p2, m, n, P, bits, L, T2, lane);
    else score = gotoh_pass_impl<K, 1>(p1, p2, m, n, P, bits, L, T2, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tl = gotoh_traceback<K>(bits, m, n, L.trace, lane, tailV, tailH);
  return score;
}

// gotoh(a1, a2, align, AlignConfig<true,true>, sc): merges two nodes into `out`
// (rows a1 then rows a2, row stride LCAP).  Returns 0, DELLYHIP_E_LIMIT, or (SLOW == false only)
// MSA_DEFER when a node has more column types than the score table holds.
template <bool SLOW>
__device__ __forceinline__ int merge_nodes(const Node& a1, const Node& a2, uint8_t* out, int& out_len, int& score,
                                           const dellyhip_params& P, uint32_t* prof, uint32_t* bits, MsaLds& L,
                                           int lane, int tmax) {
  const int m = a1.len, n = a2.len;
  if (m > LCAP - 1 || n > LCAP || m + 1 > WAVE * GKMAX) return DELLYHIP_E_LIMIT;
  uint32_t* p1 = prof;
  uint32_t* p2 = prof + (size_t)LCAP * PROFW;
  const bool single = (a1.rows == 1 && a2.rows == 1);
  int T2 = 0;
  if (!SLOW && m + 1 > WAVE * FASTK) return MSA_DEFER;
  if (single) {
    build_single(a1, p1, lane);
    build_single(a2, p2, lane);
  } else if (SLOW) {
    build_profile(a1, p1, L, lane);
    __syncthreads();
    build_profile(a2, p2, L, lane);
  } else {
    const int T1 = build_types(a1, 0, L, lane, tmax);
    if (T1 < 0) return MSA_DEFER;
    T2 = build_types(a2, 1, L, lane, tmax);
    if (T2 < 0) return MSA_DEFER;
    if (!fill_table(T1, T2, P, L, lane)) return MSA_DEFER;
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int tl = 0, tailV = 0, tailH = 0;
  const int mode = single ? 0 : (SLOW ? 2 : 1);
  const int K = (m + 1 + WAVE - 1) / WAVE;
  switch (K) {
    case 1: score = gotoh_dispatch_k<1, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH); break;
    case 2: score = gotoh_dispatch_k<2, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH); break;
    case 3: score = gotoh_dispatch_k<3, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH); break;
    case 4: score = gotoh_dispatch_k<4, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH); break;
    case 5: score = gotoh_dispatch_k<5, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH); break;
    default:
      // K = 6..8 (node longer than 319 columns) only in the direct-float kernel: their register
      // footprint would otherwise set the occupancy of the score-table kernel
      if constexpr (SLOW) {
        if (K == 6) score = gotoh_dispatch_k<6, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH);
        else if (K == 7) score = gotoh_dispatch_k<7, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH);
        else score = gotoh_dispatch_k<8, SLOW>(mode, p1, p2, m, n, P, bits, L, T2, lane, tl, tailV, tailH);
      }
      break;
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
template <bool SLOW>
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
      for (int q = lane; q < N * 5 * LCSW; q += WAVE) (&L.u.t.lcsmask[0][0][0])[q] = 0ull;
      __syncthreads();
      for (int r = 0; r < N; ++r) {
        const uint8_t* s = blob + L.roff[r];
        for (int base = 0; base < L.rlen[r]; base += WAVE) {
          int q = base + lane;
          int code = (q < L.rlen[r]) ? letter_code(s[q]) : -1;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            unsigned long long bm = __ballot(code == k);
            if (lane == 0) L.u.t.lcsmask[r][k][base >> 6] = bm;
          }
        }
      }
      const int D = NODES;
      for (int q = lane; q < D * D; q += WAVE) {
        int i = q / D, jj = q - i * D;
        L.u.t.d[q] = (jj > i) ? (int8_t)-1 : (int8_t)0;
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
          int l = lcs_bitparallel(L.u.t.lcsmask[i], blob + L.roff[i], L.rlen[i], blob + L.roff[jj], L.rlen[jj]);
          int mn = min(L.rlen[i], L.rlen[jj]);
          L.u.t.d[i * D + jj] = (int8_t)((l * 100) / mn);   // msa.h:41
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
            int dv = L.u.t.d[q];
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
            int a = (dI < i) ? L.u.t.d[dI * D + i] : L.u.t.d[i * D + dI];
            int b = (dJ < i) ? L.u.t.d[dJ * D + i] : L.u.t.d[i * D + dJ];
            L.u.t.d[i * D + nn] = (int8_t)((a + b) / 2);
          }
        }
        __syncthreads();
        for (int i = lane; i < nn + 1; i += WAVE) {
          if (i < dI) L.u.t.d[i * D + dI] = -1;
          if (i > dI) L.u.t.d[dI * D + i] = -1;
          if (i < dJ) L.u.t.d[i * D + dJ] = -1;
          if (i > dJ) L.u.t.d[dJ * D + i] = -1;
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
        int rcode = merge_nodes<SLOW>(a1, a2, alnbuf + (size_t)L.node_base[x] * LCAP, olen, score, A.p, prof, bits, L, lane,
                                      A.tmax);
        if (rcode) status = (rcode == MSA_DEFER) ? DH_DEFERRED : rcode;
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
    if (!SLOW && status == DH_DEFERRED) atomicAdd(A.defer_counter, 1);
  }
  __syncthreads();
}

#ifndef DH_MSA_WAVES
#define DH_MSA_WAVES 3
#endif
// score-table kernel: every junction; junctions with too many column types are flagged
// DH_DEFERRED for msa_slow_kernel
__global__ __launch_bounds__(WAVE, DH_MSA_WAVES) void msa_kernel(MsaArgs A, int nmax) {
  __shared__ MsaLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    msa_junction<false>(A, w, L, ws, nmax, lane);
  }
}

// direct-float kernel (per-cell profile dot product, ~250 VGPRs): only deferred junctions
__global__ __launch_bounds__(WAVE) void msa_slow_kernel(MsaArgs A, int nmax) {
  __shared__ MsaLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  if (*A.defer_counter == 0) return;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    if (A.res[w].status != DH_DEFERRED) continue;
    __syncthreads();
    msa_junction<true>(A, w, L, ws, nmax, lane);
  }
}

// single gotoh(a1, a2) on caller-supplied alignments (dellyhip_gotoh).  Runs the score-table
// path and the direct-float path and reports DELLYHIP_E_RUNTIME if they disagree (the table is
// an optimisation of the float expression, never an approximation of it).
__global__ __launch_bounds__(WAVE) void gotoh_single_kernel(MsaArgs A) {
  __shared__ MsaLds L;
  const int lane = threadIdx.x;
  Node a1{A.g_a1, A.g_r1, A.g_m, A.g_m}, a2{A.g_a2, A.g_r2, A.g_n, A.g_n};
  uint32_t* prof = reinterpret_cast<uint32_t*>(A.ws);
  uint32_t* bits = prof + 2 * (size_t)LCAP * PROFW;
  int olen = 0, score = 0, olen2 = 0, score2 = 0;
  int rc = DELLYHIP_E_LIMIT;
  if (!(A.g_r1 + A.g_r2 > 2 * NRMAX || A.g_r1 > NRMAX || A.g_r2 > NRMAX)) {
    uint8_t* out2 = A.g_out + (size_t)(A.g_r1 + A.g_r2) * LCAP;
    rc = merge_nodes<true>(a1, a2, A.g_out, olen, score, A.p, prof, bits, L, lane, A.tmax);
    const int rc2 = merge_nodes<false>(a1, a2, out2, olen2, score2, A.p, prof, bits, L, lane, A.tmax);
    if (!rc && rc2 != MSA_DEFER) {
      int bad = (rc2 != 0) || (olen2 != olen) || (score2 != score);
      if (!bad)
        for (int q = lane; q < (A.g_r1 + A.g_r2) * LCAP; q += WAVE) {
          const int col = q % LCAP;
          if (col < olen && A.g_out[q] != out2[q]) bad = 1;
        }
      if (__ballot(bad) != 0ull) rc = DELLYHIP_E_RUNTIME;
    }
    if (lane == 0) A.g_info[3] = (rc2 == MSA_DEFER) ? 0 : 1;   // 1: the table path ran
  }
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

// score-table types need |match|, |mismatch| <= 127 (int8 table); otherwise every profile merge
// goes to the direct-float kernel
inline int msa_tmax(const dellyhip_params& P, int wanted) {
  if (P.match > 127 || P.match < -127 || P.mismatch > 127 || P.mismatch < -127) return 0;
  return std::max(0, std::min(wanted, TMAXC));
}

// a.work_counter and a.defer_counter must be zeroed on the stream before the call
inline int msa_launch(const MsaArgs& a, int grid, int nmax, hipStream_t s) {
  hipLaunchKernelGGL(msa_kernel, dim3(grid), dim3(WAVE), 0, s, a, nmax);
  hipLaunchKernelGGL(msa_slow_kernel, dim3(grid), dim3(WAVE), 0, s, a, nmax);
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

inline int msa_single_gotoh(hipStream_t s, const dellyhip_params& P, int tmax, const char* a1, int r1, int m,
                            const char* a2, int r2, int n, char* out, int cap, int32_t* len, int32_t* score) {
  if (r1 < 1 || r2 < 1 || r1 > NRMAX || r2 > NRMAX || m < 0 || n < 0 || m > LCAP - 1 || n > LCAP) return DELLYHIP_E_LIMIT;
  uint8_t *d1 = nullptr, *d2 = nullptr, *dout = nullptr, *ws = nullptr;
  int* dinfo = nullptr;
  size_t wsb = MsaWs::bytes(2);
  if (hipMalloc((void**)&d1, std::max(r1 * m, 1)) != hipSuccess || hipMalloc((void**)&d2, std::max(r2 * n, 1)) != hipSuccess ||
      hipMalloc((void**)&dout, (size_t)2 * (r1 + r2) * LCAP) != hipSuccess || hipMalloc((void**)&ws, wsb) != hipSuccess ||
      hipMalloc((void**)&dinfo, 32) != hipSuccess)
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
inline int msa_single(hipStream_t s, const dellyhip_params& P, int tmax, int n_reads, const char* seq_blob,
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
      hipMalloc((void**)&dlen, 4) != hipSuccess || hipMalloc((void**)&dcnt, 8) != hipSuccess)
    return DELLYHIP_E_NOMEM;
  (void)hipMemcpy(dj, &J, sizeof J, hipMemcpyHostToDevice);
  (void)hipMemcpy(dblob, seq_blob, blob_bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(doff, seq_off, (n_reads + 1) * 8, hipMemcpyHostToDevice);
  (void)hipMemset(dres, 0, sizeof(dellyhip_result));
  (void)hipMemset(dcnt, 0, 8);
  (void)hipMemset(dlen, 0, 4);
  MsaArgs A{};
  A.junc = dj; A.seq_blob = dblob; A.seq_off = doff; A.p = P; A.res = dres; A.out_blob = dout; A.out_stride = LCAP;
  A.cons_len = dlen; A.ws = ws; A.ws_stride = wsb; A.n_work = 1; A.work_counter = dcnt;
  A.defer_counter = dcnt + 1;
  A.tmax = msa_tmax(P, tmax);
  msa_launch(A, 1, nmax, s);
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
