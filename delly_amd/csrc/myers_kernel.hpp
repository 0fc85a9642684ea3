// myers_kernel.hpp -- gfx950 device code: global (NW) edit DISTANCE with Myers' bit-vector
// recurrences, the arithmetic edlib itself uses (src/edlib.cpp:390-470 calculateBlock,
// :728-930 myersCalcEditDistanceNW); the distance is unique, so no tie-breaking is involved.
// Callers: all-pairs distances of msaEdlib (src/assemble.h:386-395), the orientation test of
// _alignConsensus (src/split.h:564-572), dellyhip_edlib_align(NW, DISTANCE).
//
// One alignment per 64-lane wavefront.  The pattern (rows) is cut into 32-row words; lane l owns
// NW consecutive words (rows 32*NW*l + 1 ...), text columns advance one per step with one
// column of skew per lane, the horizontal delta (-1/0/+1) leaving a lane's last word travels
// to the next lane by DPP -- the same systolic arrangement as the DP kernels, 32 cells per
// ~25 VALU instructions instead of one.  Rows <= 64*32*MYERS_NW = 6144.
#pragma once
#include "split_kernel.hpp"

namespace dh {

constexpr int MYERS_NW = 3;                    // 32-bit words per lane
constexpr int MYERS_ROWS = WAVE * 32 * MYERS_NW;

// equality masks of one 32-row word for the letters A, C, G, T, N; other bytes are compared
// on the fly (rare)
struct MyersWord {
  uint32_t eq[5];
};

__device__ __forceinline__ int myers_code(int c) {
  return ((unsigned)c > 255u) ? -1 : letter_code_bf((uint8_t)c);   // (NOMATCH and other non-bytes: none)
}

// equality mask of a 32-row word for a text byte outside ACGTN (rare; kept out of line so that
// the hot loop stays small enough for the instruction cache)
__device__ __noinline__ uint32_t myers_eq_slow(const uint8_t* pattern, int pn, int lo, int b) {
  uint32_t Eq = 0;
  for (int q = 0; q < 32; ++q) {
    const int r = lo + q;
    if (r < pn && (int)pattern[r] == b) Eq |= 1u << q;
  }
  return Eq;
}

// Returns the NW edit distance of pattern[0..pn) vs text[0..tn) (both > 0, pn <= 64*32*NWORDS).
// pattern / text: any address space (LDS or global); exact byte comparison.
// Strips: a pattern longer than 64*32*NWORDS rows is cut into strips that are swept one after the other over the
// whole text (myers_nw_big).  `row_base` = rows above this strip (D[row_base + i][0] = row_base + i); hin_buf[c]
// (c = 1..tn) = the horizontal delta D[row_base][c] - D[row_base][c-1] leaving the strip above (nullptr: row 0 of the
// matrix, +1 everywhere); hout_buf[c] receives the delta leaving this strip's last row (the strip must be full).
template <int NWORDS>
__device__ __noinline__ int myers_nw_distance(const uint8_t* pattern, int pn, const uint8_t* text, int tn, int lane,
                                              int row_base = 0, const int8_t* hin_buf = nullptr, int8_t* hout_buf = nullptr) {
  MyersWord W[NWORDS];
  uint32_t Pv[NWORDS], Mv[NWORDS];
  const int row0 = lane * 32 * NWORDS;   // zero-based first row of this lane
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
#pragma unroll
    for (int k = 0; k < 5; ++k) W[w].eq[k] = 0;
    for (int b = 0; b < 32; ++b) {
      const int r = row0 + w * 32 + b;
      if (r < pn) {
        const int code = myers_code((int)pattern[r]);
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (code == k) W[w].eq[k] |= 1u << b;
      }
    }
    Pv[w] = 0xffffffffu;   // D[i][0] = i
    Mv[w] = 0;
  }
  // score at the bottom row of this lane's last word, column 0
  int score = row_base + row0 + 32 * NWORDS;
  const int lastlane = (pn - 1) / (32 * NWORDS);
  const int T = tn + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 1;      // horizontal delta entering this lane (lane 0: D[0][j] - D[0][j-1] = +1)
  int b = NOMATCH;
  int c = -lane;
  int fin = 0;
  int outv = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int ci = blk * 16 + (lane & 15);
    const int chunk = (ci < tn) ? (int)text[ci] : NOMATCH;
    const int bchunk = (hin_buf && ci + 1 <= tn) ? (int)hin_buf[ci + 1] : 1;   // lane 0's column at step 16*blk + f is 16*blk + f + 1
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {   // rolled: ~60 instructions per step, the body must stay I-cache resident
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      b = dpp_from_prev(b, newc);
      const int hin0 = dpp_from_prev(hcarry, __builtin_amdgcn_readlane(bchunk, f));
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)tn) {
        const int code = myers_code(b);
        int hin = hin0;
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) {
          uint32_t Eq;
          if (code >= 0) {
            Eq = code == 0 ? W[w].eq[0] : code == 1 ? W[w].eq[1] : code == 2 ? W[w].eq[2] : code == 3 ? W[w].eq[3] : W[w].eq[4];
          } else {   // foreign byte: exact comparison against the pattern rows of this word
            Eq = myers_eq_slow(pattern, pn, row0 + w * 32, b);
          }
          // edlib.cpp:390-470 (Hyyro's block step), 32-bit words
          const uint32_t hinNeg = (hin < 0) ? 1u : 0u;
          const uint32_t Xv = Eq | Mv[w];
          Eq |= hinNeg;
          const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
          uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
          uint32_t Mh = Pv[w] & Xh;
          const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
          Ph <<= 1;
          Mh <<= 1;
          Mh |= hinNeg;
          Ph |= (hin > 0) ? 1u : 0u;
          Pv[w] = Mh | ~(Xv | Ph);
          Mv[w] = Ph & Xv;
          hin = hout;
        }
        hcarry = hin;
        score += hin;
        if (c == tn) {
          // D[pn][tn] lives in lane `lastlane`: bottom score minus the vertical deltas of the rows below pn
          const int below = row0 + 32 * NWORDS - pn;   // rows of this lane beyond the pattern
          int s = score;
          if (below > 0) {
#pragma unroll
            for (int w = 0; w < NWORDS; ++w) {
              const int lo = row0 + w * 32;             // zero-based first row of word w
              const int nb = min(32, max(0, lo + 32 - pn));   // how many of its rows are >= pn
              if (nb > 0) {
                const uint32_t mk = (nb >= 32) ? 0xffffffffu : (~0u << (32 - nb));
                s -= __popc(Pv[w] & mk);
                s += __popc(Mv[w] & mk);
              }
            }
          }
          fin = s;
        }
      }
      if (hout_buf) {   // the last lane's delta of column 16*blk + f - 62 parks in lane f
        const int sv = __builtin_amdgcn_readlane(hcarry, 63);
        outv = (lane == f) ? sv : outv;
      }
    }
    if (hout_buf) {
      const int col = blk * 16 + lane - 62;
      if (lane < 16 && col >= 1 && col <= tn) hout_buf[col] = (int8_t)outv;
    }
  }
  return __shfl(fin, lastlane);
}

__device__ __forceinline__ int myers_nw(const uint8_t* pattern, int pn, const uint8_t* text, int tn, int lane) {
  if (pn <= WAVE * 32) return myers_nw_distance<1>(pattern, pn, text, tn, lane);
  if (pn <= WAVE * 64) return myers_nw_distance<2>(pattern, pn, text, tn, lane);
  return myers_nw_distance<3>(pattern, pn, text, tn, lane);
}

// edlib HW ("infix") DISTANCE of pattern[0..pn) inside text[0..tn): min over all text end columns of the last pattern row
// with a free start (D[0][j] = 0, src/edlib.cpp:545-700); pn <= 64*32*NWORDS, both > 0.  Same layout as
// myers_nw_distance; the horizontal delta entering the first row is 0 and the bottom score is tracked per column.
template <int NWORDS>
__device__ __noinline__ int myers_hw_distance(const uint8_t* pattern, int pn, const uint8_t* text, int tn, int lane) {
  MyersWord W[NWORDS];
  uint32_t Pv[NWORDS], Mv[NWORDS], mk[NWORDS];
  const int row0 = lane * 32 * NWORDS;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
#pragma unroll
    for (int k = 0; k < 5; ++k) W[w].eq[k] = 0;
    for (int b = 0; b < 32; ++b) {
      const int r = row0 + w * 32 + b;
      if (r < pn) {
        const int code = myers_code((int)pattern[r]);
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (code == k) W[w].eq[k] |= 1u << b;
      }
    }
    Pv[w] = 0xffffffffu;   // D[i][0] = i
    Mv[w] = 0;
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, lo + 32 - pn));   // rows of this word at or beyond pn
    mk[w] = (nb >= 32) ? 0xffffffffu : ((nb > 0) ? (~0u << (32 - nb)) : 0u);
  }
  int score = row0 + 32 * NWORDS;
  const int lastlane = (pn - 1) / (32 * NWORDS);
  const int T = tn + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 0;
  int b = NOMATCH;
  int c = -lane;
  int best = pn;           // column 0: D[pn][0] = pn
  for (int blk = 0; blk < nblk; ++blk) {
    const int ci = blk * 16 + (lane & 15);
    const int chunk = (ci < tn) ? (int)text[ci] : NOMATCH;
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      b = dpp_from_prev(b, newc);
      const int hin0 = dpp_from_prev(hcarry, 0);   // row 0 is free: D[0][j] - D[0][j-1] = 0
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)tn) {
        const int code = myers_code(b);
        int hin = hin0;
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) {
          uint32_t Eq;
          if (code >= 0) Eq = code == 0 ? W[w].eq[0] : code == 1 ? W[w].eq[1] : code == 2 ? W[w].eq[2] : code == 3 ? W[w].eq[3] : W[w].eq[4];
          else Eq = myers_eq_slow(pattern, pn, row0 + w * 32, b);
          const uint32_t hinNeg = (hin < 0) ? 1u : 0u;
          const uint32_t Xv = Eq | Mv[w];
          Eq |= hinNeg;
          const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
          uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
          uint32_t Mh = Pv[w] & Xh;
          const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
          Ph <<= 1;
          Mh <<= 1;
          Mh |= hinNeg;
          Ph |= (hin > 0) ? 1u : 0u;
          Pv[w] = Mh | ~(Xv | Ph);
          Mv[w] = Ph & Xv;
          hin = hout;
        }
        hcarry = hin;
        score += hin;
        int s = score;   // D[pn][c] in the lane that owns row pn
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) s += __popc(Mv[w] & mk[w]) - __popc(Pv[w] & mk[w]);
        best = min(best, s);
      }
    }
  }
  return __shfl(best, lastlane);
}

// any pattern length: strips of MYERS_ROWS rows; hb0 / hb1: two byte arrays of tn + 16 entries (global memory)
__device__ __forceinline__ int myers_nw_big(const uint8_t* pattern, int pn, const uint8_t* text, int tn, int8_t* hb0, int8_t* hb1,
                                            int lane) {
  if (pn <= MYERS_ROWS) return myers_nw(pattern, pn, text, tn, lane);
  const int S = (pn + MYERS_ROWS - 1) / MYERS_ROWS;
  int d = 0;
  for (int q = 0; q < S; ++q) {
    const int base = q * MYERS_ROWS;
    const int8_t* hin = (q > 0) ? ((q & 1) ? hb0 : hb1) : nullptr;
    int8_t* hout = (q + 1 < S) ? ((q & 1) ? hb1 : hb0) : nullptr;
    d = myers_nw_distance<MYERS_NW>(pattern + base, min(MYERS_ROWS, pn - base), text, tn, lane, base, hin, hout);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DH_SYNC();
  }
  return d;
}

// ---- branch-free variant for the batched long-read genotyping distances ---------------------------------
// Same recurrence and layout as myers_nw_distance, but the column step has no divergent control flow: the
// per-lane equality masks sit in LDS as [word][letter A,C,G,T,N,none][lane] behind a byte -> slot table, lanes
// outside their column range keep their state through selects (so every lane ends frozen at column tn and the
// bottom-right value is read off once after the loop).  A text byte outside ACGTN takes the all-zero mask, which
// is exact as long as the pattern holds no such byte; otherwise the function declines (returns -1) and the
// caller uses myers_nw_distance.
template <int NWORDS>
struct MyersLds {
  uint16_t lut[256];                 // first: the same offset for every NWORDS (the kernel holds one MyersLds<MYERS_NW>)
  uint32_t eq[NWORDS * 6 * WAVE];
};

__device__ __forceinline__ void myers_lut_init(uint16_t* lut, int lane) {
  for (int i = lane; i < 256; i += WAVE) {
    const int code = myers_code(i);
    lut[i] = (uint16_t)((code < 0 ? 5 : code) * WAVE);
  }
}

typedef unsigned int dh_u32x4 __attribute__((ext_vector_type(4)));
typedef dh_u32x4 __attribute__((aligned(1))) dh_u32x4_unaligned;   // (the strings of a blob start anywhere)

template <int NWORDS>
__device__ __noinline__ int myers_nw_fast(MyersLds<NWORDS>& L, const uint8_t* pattern_, int pn, const uint8_t* text_, int tn,
                                             int lane) {
  // pattern / text are in HBM for every caller (the pair lists of myers_pairs_kernel / nw_jobs_kernel): accessed through
  // global-space pointers the text load of the next chunk does not hold up the LDS mask reads of the current one
  // (a flat load counts against the LDS counter too; split_kernel.hpp, gptr_cu8)
  const gptr_cu8 pattern = (gptr_cu8)pattern_, text = (gptr_cu8)text_;
  const int row0 = lane * 32 * NWORDS;
  bool foreign = false;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    uint32_t* slot = &L.eq[w * 6 * WAVE + lane];
#pragma unroll
    for (int k = 0; k < 6; ++k) slot[k * WAVE] = 0;
    const int rbeg = row0 + w * 32;
    if (rbeg < pn) {   // the word's 32 pattern bytes in two loads (the blob is padded: reading past pn is harmless)
      const dh_u32x4 g0 = *reinterpret_cast<const __attribute__((address_space(1))) dh_u32x4_unaligned*>(pattern + rbeg);
      const dh_u32x4 g1 = *reinterpret_cast<const __attribute__((address_space(1))) dh_u32x4_unaligned*>(pattern + rbeg + 16);
      const uint32_t wd[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int code = (int)L.lut[(wd[q >> 2] >> ((q & 3) * 8)) & 0xff];
        const bool in = rbeg + q < pn;
        foreign |= in && code == 5 * WAVE;
        if (in) slot[code] |= 1u << q;
      }
    }
    slot[5 * WAVE] = 0;
  }
  if (__ballot(foreign)) return -1;
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = 0xffffffffu;
    Mv[w] = 0;
  }
  int score = row0 + 32 * NWORDS;
  const int lastlane = (pn - 1) / (32 * NWORDS);
  const int T = tn + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 1;
  int c = -lane;
  // The text travels through the lanes as mask-slot offsets (byte -> slot translated once per 16-column chunk),
  // and the masks of step t+1 are fetched while the arithmetic of step t runs.
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    return (int)L.lut[(ci < tn) ? (int)text[ci] : 0];
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));   // slot offset of this lane's next column
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = L.eq[w * 6 * WAVE + bs + lane];
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
      // next column's slot and masks
      const int newc = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      bs = dpp_from_prev(bs, newc);
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqN[w] = L.eq[w * 6 * WAVE + bs + lane];
      int hin = dpp_from_prev(hcarry, 1);
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)tn;
      uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        uint32_t Eq = EqC[w];
        const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470 (Hyyro's block step), 32-bit words
        const uint32_t Xv = Eq | Mv[w];
        Eq |= hinNeg;
        const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
        uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
        uint32_t Mh = Pv[w] & Xh;
        const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1u : 0u;
        nP[w] = Mh | ~(Xv | Ph);
        nM[w] = Ph & Xv;
        hin = hout;
      }
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        Pv[w] = valid ? nP[w] : Pv[w];
        Mv[w] = valid ? nM[w] : Mv[w];
      }
      hcarry = valid ? hin : hcarry;
      score += valid ? hin : 0;
    }
    chunk = chunk_next;
  }
  // every lane is frozen at column tn: D[pn][tn] = bottom score of lane `lastlane` minus the vertical deltas below row pn
  int s = score;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, lo + 32 - pn));
    if (nb > 0) {
      const uint32_t mk = (nb >= 32) ? 0xffffffffu : (~0u << (32 - nb));
      s -= __popc(Pv[w] & mk);
      s += __popc(Mv[w] & mk);
    }
  }
  return __shfl(s, lastlane);
}

// TWO independent pairs per wavefront, one per 32-lane half (lanes 0-31: pair A, lanes 32-63: pair B): a 2.3 kb read is 72
// words of 32 rows -- with one pair per wavefront that takes two words per lane on all 64 lanes (44 % padding), with
// two pairs three words per lane on 32 lanes each (25 % padding) and the step costs ~1.3x.  Same recurrence and
// staging as myers_nw_fast; lane 32 starts pair B's carry / text chain instead of continuing lane 31's.  Patterns of up
// to 32 * 32 * NWORDS rows.  Returns false when a pattern holds bytes outside ACGTN (the caller falls back, per pair).
template <int NWORDS>
__device__ __noinline__ bool myers_nw_fast_x2(MyersLds<NWORDS>& L, const uint8_t* patA, int pnA, const uint8_t* txtA, int tnA,
                                              const uint8_t* patB, int pnB, const uint8_t* txtB, int tnB, int lane, int& dA, int& dB) {
  const bool hiHalf = lane >= 32;
  const int hl = lane & 31;
  const gptr_cu8 pattern = (gptr_cu8)(hiHalf ? patB : patA);   // (HBM for every caller: see myers_nw_fast)
  const gptr_cu8 text = (gptr_cu8)(hiHalf ? txtB : txtA);
  const int pn = hiHalf ? pnB : pnA, tn = hiHalf ? tnB : tnA;
  const int row0 = hl * 32 * NWORDS;
  bool foreign = false;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    uint32_t* slot = &L.eq[w * 6 * WAVE + lane];
#pragma unroll
    for (int k = 0; k < 6; ++k) slot[k * WAVE] = 0;
    const int rbeg = row0 + w * 32;
    if (rbeg < pn) {
      const dh_u32x4 g0 = *reinterpret_cast<const __attribute__((address_space(1))) dh_u32x4_unaligned*>(pattern + rbeg);
      const dh_u32x4 g1 = *reinterpret_cast<const __attribute__((address_space(1))) dh_u32x4_unaligned*>(pattern + rbeg + 16);
      const uint32_t wd[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int code = (int)L.lut[(wd[q >> 2] >> ((q & 3) * 8)) & 0xff];
        const bool in = rbeg + q < pn;
        foreign |= in && code == 5 * WAVE;
        if (in) slot[code] |= 1u << q;
      }
    }
    slot[5 * WAVE] = 0;
  }
  if (__ballot(foreign)) return false;
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = 0xffffffffu;
    Mv[w] = 0;
  }
  int score = row0 + 32 * NWORDS;
  const int lastA = (pnA - 1) / (32 * NWORDS), lastB = (pnB - 1) / (32 * NWORDS);
  const int T = max(tnA + lastA, tnB + lastB);
  const int nblk = (T + 15) >> 4;
  const bool first_of_b = lane == 32;
  int hcarry = 1;
  int c = -hl;
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    return (int)L.lut[(ci < tn) ? (int)text[ci] : 0];
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));
  bs = first_of_b ? __builtin_amdgcn_readlane(chunk, 32) : bs;
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = L.eq[w * 6 * WAVE + bs + lane];
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
      const int newA = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      const int newB = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 32) : __builtin_amdgcn_readlane(chunk, 32 + f + 1);
      bs = dpp_from_prev(bs, newA);
      bs = first_of_b ? newB : bs;
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqN[w] = L.eq[w * 6 * WAVE + bs + lane];
      int hin = dpp_from_prev(hcarry, 1);
      hin = first_of_b ? 1 : hin;
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)tn;
      uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        uint32_t Eq = EqC[w];
        const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470 (Hyyro's block step), 32-bit words
        const uint32_t Xv = Eq | Mv[w];
        Eq |= hinNeg;
        const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
        uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
        uint32_t Mh = Pv[w] & Xh;
        const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1u : 0u;
        nP[w] = Mh | ~(Xv | Ph);
        nM[w] = Ph & Xv;
        hin = hout;
      }
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        Pv[w] = valid ? nP[w] : Pv[w];
        Mv[w] = valid ? nM[w] : Mv[w];
      }
      hcarry = valid ? hin : hcarry;
      score += valid ? hin : 0;
    }
    chunk = chunk_next;
  }
  int s = score;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, lo + 32 - pn));
    if (nb > 0) {
      const uint32_t mk = (nb >= 32) ? 0xffffffffu : (~0u << (32 - nb));
      s -= __popc(Pv[w] & mk);
      s += __popc(Mv[w] & mk);
    }
  }
  dA = __shfl(s, lastA);
  dB = __shfl(s, 32 + lastB);
  return true;
}
// rows one half-wavefront holds
constexpr int MYERS_HALF_ROWS = 32 * 32 * MYERS_NW;
// side by side pays when the shared step (words per lane for the longer pattern on 32 lanes) is cheaper than the two
// separate passes (words per lane on 64 lanes each): 2.3 kb reads 3 < 2 + 2, 1.5 kb strings 2 = 1 + 1 (measured: no gain)
__host__ __device__ __forceinline__ bool myers_x2_pays(int pn0, int pn1) {
  if (pn0 < 1 || pn1 < 1 || pn0 > MYERS_HALF_ROWS || pn1 > MYERS_HALF_ROWS) return false;
  const int both = ((pn0 > pn1 ? pn0 : pn1) + 1023) >> 10;
  const int apart = ((pn0 + 2047) >> 11) + ((pn1 + 2047) >> 11);
  return both < apart;
}
__device__ __forceinline__ bool myers_nw_auto_x2(MyersLds<MYERS_NW>& L, const uint8_t* patA, int pnA, const uint8_t* txtA, int tnA,
                                                 const uint8_t* patB, int pnB, const uint8_t* txtB, int tnB, int lane, int& dA, int& dB) {
  const int pmax = max(pnA, pnB);
  if (pmax <= 32 * 32) return myers_nw_fast_x2<1>(reinterpret_cast<MyersLds<1>&>(L), patA, pnA, txtA, tnA, patB, pnB, txtB, tnB, lane, dA, dB);
  if (pmax <= 32 * 64) return myers_nw_fast_x2<2>(reinterpret_cast<MyersLds<2>&>(L), patA, pnA, txtA, tnA, patB, pnB, txtB, tnB, lane, dA, dB);
  return myers_nw_fast_x2<3>(L, patA, pnA, txtA, tnA, patB, pnB, txtB, tnB, lane, dA, dB);
}

// Two patterns of the same length against one text in lock-step (the orientation test of _alignConsensus: consensus and
// its reverse complement vs the window, src/split.h:564-572): one pass over the text, twice the independent work per
// step.  L2: a second mask area for pattern B.  Returns false when a pattern holds bytes outside ACGTN (caller falls back).
template <int NWORDS>
__device__ __noinline__ bool myers_nw_fast2(MyersLds<NWORDS>& L, uint32_t* eqB, const uint8_t* patA, const uint8_t* patB, int pn,
                                            const uint8_t* text_, int tn, int lane, int& dA, int& dB) {
  // (the text lives in the wavefront's HBM workspace for every caller: through a global-space pointer its chunk loads are
  //  global_load, not flat_load -- a flat load counts against the LDS counter too, and the mask reads of the next sixteen steps
  //  waited for the chunk's HBM round trip; same remedy as in myers_nw_fast, round 6)
  const gptr_cu8 text = (gptr_cu8)text_;
  const int row0 = lane * 32 * NWORDS;
  bool foreign = false;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint8_t* pattern = p ? patB : patA;
    uint32_t* eqp = p ? eqB : L.eq;
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) {
      uint32_t* slot = &eqp[w * 6 * WAVE + lane];
#pragma unroll
      for (int k = 0; k < 6; ++k) slot[k * WAVE] = 0;
      const int rbeg = row0 + w * 32;
      if (rbeg < pn) {
        uint4 v[2];
        __builtin_memcpy(&v[0], pattern + rbeg, 16);
        __builtin_memcpy(&v[1], pattern + rbeg + 16, 16);
        const uint32_t wd[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int code = (int)L.lut[(wd[q >> 2] >> ((q & 3) * 8)) & 0xff];
          const bool in = rbeg + q < pn;
          foreign |= in && code == 5 * WAVE;
          if (in) slot[code] |= 1u << q;
        }
      }
      slot[5 * WAVE] = 0;
    }
  }
  if (__ballot(foreign)) return false;
  uint32_t Pv[2][NWORDS], Mv[2][NWORDS];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) {
      Pv[p][w] = 0xffffffffu;
      Mv[p][w] = 0;
    }
  int score[2] = {row0 + 32 * NWORDS, row0 + 32 * NWORDS};
  const int lastlane = (pn - 1) / (32 * NWORDS);
  const int T = tn + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry[2] = {1, 1};
  int c = -lane;
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    return (int)L.lut[(ci < tn) ? (int)text[ci] : 0];
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));
  uint32_t EqN[2][NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    EqN[0][w] = L.eq[w * 6 * WAVE + bs + lane];
    EqN[1][w] = eqB[w * 6 * WAVE + bs + lane];
  }
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[2][NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        EqC[0][w] = EqN[0][w];
        EqC[1][w] = EqN[1][w];
      }
      const int newc = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      bs = dpp_from_prev(bs, newc);
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        EqN[0][w] = L.eq[w * 6 * WAVE + bs + lane];
        EqN[1][w] = eqB[w * 6 * WAVE + bs + lane];
      }
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)tn;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        int hin = dpp_from_prev(hcarry[p], 1);
        uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) {
          uint32_t Eq = EqC[p][w];
          const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470 (Hyyro's block step), 32-bit words
          const uint32_t Xv = Eq | Mv[p][w];
          Eq |= hinNeg;
          const uint32_t Xh = (((Eq & Pv[p][w]) + Pv[p][w]) ^ Pv[p][w]) | Eq;
          uint32_t Ph = Mv[p][w] | ~(Xh | Pv[p][w]);
          uint32_t Mh = Pv[p][w] & Xh;
          const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
          Ph <<= 1;
          Mh <<= 1;
          Mh |= hinNeg;
          Ph |= (hin > 0) ? 1u : 0u;
          nP[w] = Mh | ~(Xv | Ph);
          nM[w] = Ph & Xv;
          hin = hout;
        }
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) {
          Pv[p][w] = valid ? nP[w] : Pv[p][w];
          Mv[p][w] = valid ? nM[w] : Mv[p][w];
        }
        hcarry[p] = valid ? hin : hcarry[p];
        score[p] += valid ? hin : 0;
      }
    }
    chunk = chunk_next;
  }
  int out[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int s = score[p];
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) {
      const int lo = row0 + w * 32;
      const int nb = min(32, max(0, lo + 32 - pn));
      if (nb > 0) {
        const uint32_t mk = (nb >= 32) ? 0xffffffffu : (~0u << (32 - nb));
        s -= __popc(Pv[p][w] & mk);
        s += __popc(Mv[p][w] & mk);
      }
    }
    out[p] = __shfl(s, lastlane);
  }
  dA = out[0];
  dB = out[1];
  return true;
}

// fast variant with the exact-compare fallback; L: one MyersLds<MYERS_NW> per wavefront, lut initialised
__device__ __forceinline__ int myers_nw_auto(MyersLds<MYERS_NW>& L, const uint8_t* pattern, int pn, const uint8_t* text, int tn,
                                             int lane) {
  int d;
  if (pn <= WAVE * 32) d = myers_nw_fast<1>(reinterpret_cast<MyersLds<1>&>(L), pattern, pn, text, tn, lane);
  else if (pn <= WAVE * 64) d = myers_nw_fast<2>(reinterpret_cast<MyersLds<2>&>(L), pattern, pn, text, tn, lane);
  else d = myers_nw_fast<3>(L, pattern, pn, text, tn, lane);
  if (d < 0) d = myers_nw(pattern, pn, text, tn, lane);   // pattern with bytes outside ACGTN
  return d;
}

// ---- single call (dellyhip_edlib_align, NW + DISTANCE beyond the insertion-kernel shapes) ----
__global__ __launch_bounds__(WAVE) void myers_single_kernel(const uint8_t* q, int qn, const uint8_t* t, int tn, int32_t* out) {
  const int lane = threadIdx.x;
  // rows = the shorter of the two strings would be cheaper; edlib's distance is symmetric
  const int d = myers_nw(q, qn, t, tn, lane);
  if (lane == 0) out[0] = d;
}

}  // namespace dh
#include "myers_band.hpp"
namespace dh {
#ifdef DH_LR_TIMING
__device__ unsigned long long dh_lrt_pairs[4];   // profiling builds: [0] pairs through the banded passes, [1] of them not certified (full pass)
#endif


// ---- all pairs of a junction's reads: msaEdlib's distance matrix (src/assemble.h:386-395) ----
struct PairArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  const int32_t* pair_first;   // first work item of junction j (prefix sums of n(n-1)/2), n_junc + 1 entries
  int32_t n_junc;
  int32_t n_items;
  int32_t nrmax;               // row stride of a junction's matrix
  int32_t* edit;               // edit[j*nrmax*nrmax + a*nrmax + b]
  int8_t* hbuf;                // per block: 2 x hbuf_half bytes for the strip passes of pairs with both reads > MYERS_ROWS
  uint64_t hbuf_half;          // (0: no such pair in the batch)
  int32_t band_g, band_k, band_wl;   // banded passes (myers_band.hpp): pairs per wavefront (0: off), band, lanes per pair -- from the batch's longest read
};

__global__ __launch_bounds__(WAVE) void myers_pairs_kernel(PairArgs A) {
  MyersBandLds& LB = myers_band_lds();
  MyersLds<MYERS_NW>& L = LB.full();
  const int lane = threadIdx.x;
  myers_lut_init(L.lut, lane);
  DH_SYNC();
  struct Item { int j, a, bb, la, lb; uint64_t oa, ob; };
  auto describe = [&](int item) -> Item {
    // junction of this item: binary search in pair_first
    int lo = 0, hi = A.n_junc;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (A.pair_first[mid] <= item) lo = mid;
      else hi = mid;
    }
    Item I;
    I.j = lo;
    const dellyhip_junction J = A.junc[I.j];
    const int N = J.n_seq;
    int rem = item - A.pair_first[I.j], a = 0;
    while (rem >= N - 1 - a) {
      rem -= N - 1 - a;
      ++a;
    }
    I.a = a;
    I.bb = a + 1 + rem;
    I.oa = A.seq_off[J.seq_first + I.a];
    I.ob = A.seq_off[J.seq_first + I.bb];
    I.la = (int)(A.seq_off[J.seq_first + I.a + 1] - I.oa);
    I.lb = (int)(A.seq_off[J.seq_first + I.bb + 1] - I.ob);
    return I;
  };
  auto store = [&](const Item& I, int d) {
    if (lane == 0) {
      int32_t* E = A.edit + (size_t)I.j * A.nrmax * A.nrmax;
      E[I.a * A.nrmax + I.bb] = d;
      E[I.bb * A.nrmax + I.a] = d;
    }
  };
  auto single = [&](const Item& I) {
    const int la = I.la, lb = I.lb;
    const uint64_t oa = I.oa, ob = I.ob;
    int d;
    if (la == 0 || lb == 0) d = max(la, lb);                 // edlib.cpp:160-166
    else if (la > MYERS_ROWS && lb > MYERS_ROWS) {           // strips of 6144 rows (pattern = the shorter read)
      int8_t* hb = A.hbuf + (size_t)blockIdx.x * 2 * A.hbuf_half;
      if (!A.hbuf || (uint64_t)max(la, lb) + 16 > A.hbuf_half) d = -1;  // (cannot happen: the host sizes hbuf from the longest read)
      else if (la <= lb) d = myers_nw_big(A.seq_blob + oa, la, A.seq_blob + ob, lb, hb, hb + A.hbuf_half, lane);
      else d = myers_nw_big(A.seq_blob + ob, lb, A.seq_blob + oa, la, hb, hb + A.hbuf_half, lane);
    }
    else if (la <= lb || lb > MYERS_ROWS) d = myers_nw_auto(L, A.seq_blob + oa, la, A.seq_blob + ob, lb, lane);   // (the blob is padded)
    else d = myers_nw_auto(L, A.seq_blob + ob, lb, A.seq_blob + oa, la, lane);
    store(I, d);
  };
  // banded passes, band_g items per wavefront and step (myers_band.hpp); what the band does not certify falls back below
  if (A.band_g >= 2) {
    __shared__ Item its[MB_G];
    const int G = A.band_g;
    const int ngroups = (A.n_items + G - 1) / G;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      const int first = grp * G, cnt = min(G, A.n_items - first);
      MbLds& M = LB.band();
      for (int q = 0; q < cnt; ++q) {
        const Item I = describe(first + q);
        if (lane == 0) {
          its[q] = I;
          const bool ab = I.la <= I.lb;   // pattern = the shorter read (the distance is symmetric)
          M.item[q].pat = A.seq_blob + (ab ? I.oa : I.ob);
          M.item[q].txt = A.seq_blob + (ab ? I.ob : I.oa);
          M.item[q].pn = min(I.la, I.lb);
          M.item[q].tn = max(I.la, I.lb);
        }
      }
      DH_SYNC();
      myers_band_multi(cnt, A.band_k, A.band_wl, lane);
      int res[MB_G];
#pragma unroll
      for (int q = 0; q < MB_G; ++q) res[q] = (q < cnt) ? M.res[q] : 0;
      DH_SYNC();
#pragma unroll
      for (int q = 0; q < MB_G; ++q) {
        if (q < cnt) {
          const Item I = its[q];
#ifdef DH_LR_TIMING
          if (lane == 0) { atomicAdd(&dh_lrt_pairs[0], 1ull); if (res[q] < 0) atomicAdd(&dh_lrt_pairs[1], 1ull); }
#endif
          if (rfl(res[q]) >= 0) store(I, rfl(res[q]));
          else single(I);     // (rewrites the tables' bytes: the results are in registers)
        }
      }
      DH_SYNC();
    }
    return;
  }
  // items 2k and 2k + 1 side by side in the two halves of the wavefront when both shorter reads fit 3072 rows
  const int n_pairs_of_items = (A.n_items + 1) >> 1;
  for (int k = blockIdx.x; k < n_pairs_of_items; k += gridDim.x) {
    const Item I0 = describe(2 * k);
    const bool two = 2 * k + 1 < A.n_items;
    bool done = false;
    if (two) {
      const Item I1 = describe(2 * k + 1);
      const int pn0 = min(I0.la, I0.lb), pn1 = min(I1.la, I1.lb);
      if (myers_x2_pays(pn0, pn1)) {
        const uint8_t* a0 = A.seq_blob + I0.oa; const uint8_t* b0 = A.seq_blob + I0.ob;
        const uint8_t* a1 = A.seq_blob + I1.oa; const uint8_t* b1 = A.seq_blob + I1.ob;
        int d0 = 0, d1 = 0;
        done = myers_nw_auto_x2(L, (I0.la <= I0.lb) ? a0 : b0, pn0, (I0.la <= I0.lb) ? b0 : a0, max(I0.la, I0.lb),
                                (I1.la <= I1.lb) ? a1 : b1, pn1, (I1.la <= I1.lb) ? b1 : a1, max(I1.la, I1.lb), lane, d0, d1);
        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
        if (done) { store(I0, d0); store(I1, d1); }
      }
      if (!done) { single(I0); single(I1); }
    } else {
      single(I0);
    }
  }
}

// ---- a batch of independent pairs: _editDistanceNW of the long-read genotyper (src/genotype.h:21-30,276,284) ----
struct NwArgs {
  const dellyhip_nw_job* jobs;
  const uint8_t* blob;
  int32_t* dist;
  uint64_t n_jobs;
  uint32_t* next;   // work counter, zeroed before the launch: wavefronts pull job indices (pair lengths vary)
  int8_t* hbuf;     // per block: 2 x hbuf_half bytes for pairs with both strings > MYERS_ROWS (strip passes); 0 = none
  uint64_t hbuf_half;
  int32_t pairwise; // 1: wavefronts pull two adjacent jobs at a time and run them side by side where that pays (host's call)
};

__global__ __launch_bounds__(WAVE) void nw_jobs_kernel(NwArgs A) {
  __shared__ MyersLds<MYERS_NW> L;   // (the one- and two-word variants use a prefix of it)
  const int lane = threadIdx.x;
  myers_lut_init(L.lut, lane);
  DH_SYNC();
  // (the loop condition must be a scalar compare: hipcc's exec-mask structurisation of a `break` it cannot prove
  //  uniform produced a non-terminating loop here, see split_main.hpp JCtx)
  const int n_items = __builtin_amdgcn_readfirstlane((int)(uint32_t)A.n_jobs);
  auto fetch = [&]() -> int {
    int v = 0;
    if (lane == 0) v = (int)atomicAdd(A.next, A.pairwise ? 2u : 1u);
    return __builtin_amdgcn_readfirstlane(v);
  };
  auto single = [&](int item) {
    const dellyhip_nw_job J = A.jobs[item];
    const int la = (int)J.query_len, lb = (int)J.target_len;
    const uint8_t* a = A.blob + J.query_off;
    const uint8_t* b = A.blob + J.target_off;
    int d;
    if (la == 0 || lb == 0) d = max(la, lb);                              // edlib.cpp:157-163
    else if (la > MYERS_ROWS && lb > MYERS_ROWS) {
      int8_t* hb = A.hbuf + (size_t)blockIdx.x * 2 * A.hbuf_half;
      if (!A.hbuf || (uint64_t)max(la, lb) + 16 > A.hbuf_half) d = DELLYHIP_E_LIMIT;
      else if (la <= lb) d = myers_nw_big(a, la, b, lb, hb, hb + A.hbuf_half, lane);
      else d = myers_nw_big(b, lb, a, la, hb, hb + A.hbuf_half, lane);
    } else {
      const uint8_t* pat = (la <= lb) ? a : b;                             // pattern = the shorter string (symmetric)
      const uint8_t* txt = (la <= lb) ? b : a;
      const int pn = min(la, lb), tn = max(la, lb);
      d = myers_nw_auto(L, pat, pn, txt, tn, lane);
    }
    if (lane == 0) A.dist[item] = d;
  };
  // two jobs per fetch: when both patterns fit a half-wavefront they run side by side (myers_nw_fast_x2)
  for (int item = fetch(); item < n_items; item = fetch()) {
    if (!A.pairwise) { single(item); continue; }
    bool done = false;
    if (item + 1 < n_items) {
      const dellyhip_nw_job J0 = A.jobs[item], J1 = A.jobs[item + 1];
      const int la0 = (int)J0.query_len, lb0 = (int)J0.target_len, la1 = (int)J1.query_len, lb1 = (int)J1.target_len;
      const int pn0 = min(la0, lb0), pn1 = min(la1, lb1);
      if (myers_x2_pays(pn0, pn1)) {
        const uint8_t* a0 = A.blob + J0.query_off; const uint8_t* b0 = A.blob + J0.target_off;
        const uint8_t* a1 = A.blob + J1.query_off; const uint8_t* b1 = A.blob + J1.target_off;
        int d0 = 0, d1 = 0;
        done = myers_nw_auto_x2(L, (la0 <= lb0) ? a0 : b0, pn0, (la0 <= lb0) ? b0 : a0, max(la0, lb0),
                                (la1 <= lb1) ? a1 : b1, pn1, (la1 <= lb1) ? b1 : a1, max(la1, lb1), lane, d0, d1);
        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
        if (done && lane == 0) { A.dist[item] = d0; A.dist[item + 1] = d1; }
      }
    }
    if (!done) {
      single(item);
      if (item + 1 < n_items) single(item + 1);
    }
  }
}

}  // namespace dh
