// lr_kernel.hpp -- gfx950 device code for alignConsensus() on long-read shapes
// (BASELINE config C4: |consensus| ~ 2 kb, |svRefStr| ~ 7 kb; src/tegua.h:237-241 parameters):
//   optional orientation test  src/split.h:564-572  (2 x edlib NW distance)
//   longNeedle                 src/needle.h:45-222
//   _findSplit ... alleles     shared stage of split_main.hpp
//
// Same formulation as the short-read kernels (split_kernel.hpp): one junction per 64-lane
// wavefront, lanes own K = 5 consecutive DP rows, anti-diagonal skew, DPP hand-off, V' = score +
// row.  A consensus longer than 319 does not fit one pass, so the rows are cut into STRIPS of
// 64*K = 320 rows which are swept one after the other over all columns; the last row of a
// strip is written to a boundary array in global memory (one coalesced 64-byte store per 16
// steps: v_readlane of the last lane + v_writelane into a staging register) and fed to the
// first lane of the next strip (one coalesced load per 16 steps, v_readlane per step).
// Strings, boundary rows, running-max codes, direction codes and traceback ops live in a
// per-wavefront workspace in HBM (L2-resident for C4 shapes); column masks stay in LDS.
//
// Row <-> slot mapping.  The reverse-complement pass (R) and the forward pass with join (M)
// must address the same 2-bit code words: rev row rho sits at global slot g = rho + pad,
// pad = Q*320 - (m+1), strip g / 320, local slot g % 320 -- i.e. the LAST strip is full and ends
// at rev row m, the dummy slots (rho < 0) are above row 0 of the first strip, where they stay 0
// like the free first row itself.  The M pass mirrors the slots inside a strip (lane 63 leads)
// and visits the strips in reverse order, so M row r = m - rho pops exactly what R pushed.
#pragma once
#include "myers_kernel.hpp"
#include "sparse_needle.hpp"
#include "split_main.hpp"

namespace dh {

constexpr int LRK = 5;
constexpr int LRS = WAVE * LRK;                 // rows per strip
constexpr int LR_QMAX = 40;                     // strips
constexpr int LR_MMAX = LRS * LR_QMAX - 1;      // 12799 (BASELINE's stress shape: 10 kb consensus x 20 kb window)
constexpr int LR_NMAX = 32000;                  // columns < 2^15 (join key), V' = score + row <= 2m < 2^16 (scaled by 2^15 in an int)
constexpr int LR_CSHIFT = 15;                   // M pass runs on scores << 15; key = (sum' << 15) | (CINV - col)
constexpr int LR_CINV = (1 << LR_CSHIFT) - 1;
constexpr int LR_MASKW = (LR_MMAX + LR_NMAX + 127) / 64;

struct LrArgs {
  uint8_t* ws;            // per resident block
  uint64_t ws_stride;
  int32_t mcap, ncap;     // string capacities of this batch
  uint64_t off_rcons, off_ref, off_rref, off_bnd0, off_bnd1, off_br, off_trF, off_trR, off_stack;  // cons at 0
  uint64_t strip_words;   // code words per strip
  int32_t realign;        // src/split.h:564-572
  uint64_t off_masks;     // column masks of alignments beyond LR_MASKW_LDS * 64 columns (lr_masks_bytes())
  uint64_t off_sparse;    // furthest-reaching tables of the sparse longNeedle (sparse_needle.hpp)
  uint64_t sparse_bytes;  // 0: dense strip passes only
  int32_t sparse_cost;    // predicted deficit beyond which the dense strips are taken (SparseWs::pred_cap)
};

struct StrPtr {           // the four strings of a junction (workspace)
  uint8_t* cons;
  uint8_t* rcons;
  uint8_t* ref;
  uint8_t* rref;
};

struct __attribute__((aligned(16))) PostLR {
  unsigned long long mV[LR_MASKW], mR[LR_MASKW], mE[LR_MASKW];
  int32_t cumV[LR_MASKW + 1], cumR[LR_MASKW + 1];
};
// the strip kernel's LDS: the phases of a junction use it one after the other -- orientation test (bit-vector masks),
// sparse longNeedle (level tiles), column masks of the result
// column masks of lr_kernel: alignments of up to LR_MASKW_LDS * 64 columns (C4 shapes: ~10 k) keep them in LDS, longer ones
// in the wavefront's workspace -- the full-size PostLR cost 22 KB of LDS per block (6 resident blocks per CU instead of 8)
constexpr int LR_MASKW_LDS = 288;
struct __attribute__((aligned(16))) PostLRS {
  unsigned long long mV[LR_MASKW_LDS], mR[LR_MASKW_LDS], mE[LR_MASKW_LDS];
  int32_t cumV[LR_MASKW_LDS + 1], cumR[LR_MASKW_LDS + 1];
};
struct PostRef {
  unsigned long long *mV, *mR, *mE;
  int32_t *cumV, *cumR;
};
__host__ __device__ inline uint64_t lr_masks_bytes() { return 3ull * LR_MASKW * 8 + 2ull * (LR_MASKW + 1) * 4; }
struct __attribute__((aligned(16))) LrLds {
  union {
    PostLRS post;
    SpTile tile;
    struct {
      MyersLds<MYERS_NW> myers;
      uint32_t eqB[MYERS_NW * 6 * WAVE];   // masks of the second pattern of the orientation test (myers_nw_fast2)
    } o;
  } u;
  int16_t reachF[SP_LEVELS_MAX], reachR[SP_LEVELS_MAX];   // sparse longNeedle: furthest row per deficit level
};

// host + device: words of one strip's code stack
__host__ __device__ inline uint64_t lr_strip_words(int ncap) { return (uint64_t)((ncap + 63 + 15) / 16 + 1) * LRK * WAVE; }

// v_writelane_b32 with a compile-time lane index (clang has no builtin for it): lane LANE of
// `old` := the wave-uniform value `sval`
template <int LANE>
__device__ __forceinline__ int writelane_c(int sval, int old) {
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(LANE));
  return old;
}
__device__ __forceinline__ int writelane16(int sval, int lane_idx, int old) {  // lane_idx folds after unrolling
  switch (lane_idx) {
    case 0: return writelane_c<0>(sval, old);
    case 1: return writelane_c<1>(sval, old);
    case 2: return writelane_c<2>(sval, old);
    case 3: return writelane_c<3>(sval, old);
    case 4: return writelane_c<4>(sval, old);
    case 5: return writelane_c<5>(sval, old);
    case 6: return writelane_c<6>(sval, old);
    case 7: return writelane_c<7>(sval, old);
    case 8: return writelane_c<8>(sval, old);
    case 9: return writelane_c<9>(sval, old);
    case 10: return writelane_c<10>(sval, old);
    case 11: return writelane_c<11>(sval, old);
    case 12: return writelane_c<12>(sval, old);
    case 13: return writelane_c<13>(sval, old);
    case 14: return writelane_c<14>(sval, old);
    default: return writelane_c<15>(sval, old);
  }
}

// ---- strip passes ------------------------------------------------------------------------

// R pass of strip q (rev rows rho = q*320 + ls - pad).  bin/bout: boundary rows (V' of the row
// above the strip / of the strip's last row), index = column.  Pushes the running-max codes of
// pass_R (split_kernel.hpp) to `stack`, the final running maxima to brout[q*320 + ls].
// Returns the final V' of the strip's last slot (lane 63).
__device__ __noinline__ int lr_pass_R(const uint8_t* rcons, const uint8_t* rref, int m, int n, int q, int pad,
                                      const int32_t* bin, int32_t* bout, uint32_t* stack, int32_t* brout, int lane) {
  constexpr int K = LRK;
  int a[K], hg[K], h[K], br[K];
  uint32_t acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int rho = q * LRS + lane * K + i - pad;
    a[i] = (rho >= 1 && rho <= m) ? (int)rcons[rho - 1] : NOMATCH;
    hg[i] = (rho >= 1 && rho < m) ? -1 : 0;
    h[i] = 0;
    br[i] = 0;
    acc[i] = 0;
  }
  const int T = n + 63;
  const int nblk = (T + 15) >> 4;
  int upPrev = bin ? 0 : NEGBIG;   // V'[row above][0] = 0 (only lane 0 ever uses the initial value)
  int b = NOMATCH;
  int c = -lane;
  int outv = 0;
  // the column letters / boundary values of block blk+1 are loaded while block blk computes
  // (one wavefront per SIMD cannot hide a dependent HBM/L2 load per 16 steps)
  auto ld_chunk = [&](int blk) { const int ci = blk * 16 + (lane & 15); return (ci < n) ? (int)rref[ci] : NOMATCH; };
  auto ld_bnd = [&](int blk) { const int ci = blk * 16 + (lane & 15); return (bin && ci + 1 <= n) ? bin[ci + 1] : NEGBIG; };
  int chunk = ld_chunk(0), bchunk = ld_bnd(0);   // column of lane 0 at step 16*blk+f is 16*blk+f+1
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_n = ld_chunk(blk + 1), bchunk_n = ld_bnd(blk + 1);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      const int bnd = __builtin_amdgcn_readlane(bchunk, f);
      b = dpp_from_prev(b, newc);
      const int recv = dpp_from_prev(h[K - 1], bnd);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)n) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const int x = diag + ((a[i] == b) ? 2 : 0);
          const int z = h[i] + hg[i];
          const int nv = max3i(x, up, z);
          diag = h[i];
          up = nv;
          h[i] = nv;
          const int d = nv - br[i];
          br[i] = max(br[i], nv);
          const int dm = max(d, -1);
          acc[i] = acc[i] + ((uint32_t)dm << (2 * f));
        }
      }
      upPrev = recv;
      if (bout) outv = writelane16(__builtin_amdgcn_readlane(h[K - 1], 63), f, outv);
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      stack[((size_t)blk * K + i) * WAVE + lane] = acc[i] + 0x55555555u;
      acc[i] = 0;
    }
    if (bout) {   // lane f holds the last row's value of column 16*blk + f - 62
      const int col = blk * 16 + lane - 62;
      if (lane < 16 && col >= 0 && col <= n) bout[col] = outv;
    }
    chunk = chunk_n;
    bchunk = bchunk_n;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) brout[q * LRS + lane * K + i] = br[i];
  return __builtin_amdgcn_readlane(h[K - 1], 63);
}

// M pass (forward matrix + join) of the strip that mirrors R strip q.  brin: R's final running
// maxima.  Returns the strip's best join key ((sum' << 32) | (global slot << 15) | (CINV - col));
// hpad = final V' (scaled) of local slot `pad_ls` (M row m lives there in strip 0).
__device__ __noinline__ long long lr_pass_M(const uint8_t* cons, const uint8_t* ref, int m, int n, int q, int pad,
                                            const int32_t* bin, int32_t* bout, const uint32_t* stack,
                                            const int32_t* brin, int pad_ls, int lane, int& hpad) {
  constexpr int K = LRK;
  int a[K], hg[K], h[K], bm[K], g[K], bestkey[K];
  uint32_t dw[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int rho = q * LRS + lane * K + i - pad;
    const int r = m - rho;
    const bool real = (rho >= 0) && (r >= 0);
    a[i] = (real && r >= 1) ? (int)cons[r - 1] : NOMATCH;
    hg[i] = (real && r >= 1 && r < m) ? -(1 << LR_CSHIFT) : 0;
    h[i] = 0;
    bm[i] = 0;
    g[i] = real ? (brin[q * LRS + lane * K + i] << LR_CSHIFT) : NEGBIG;
    bestkey[i] = real ? (bm[i] + g[i] + LR_CINV) : (int)0x80000000;   // column 0 candidate
    dw[i] = 0;
  }
  const int T = n + 63;
  const int nblk = (T + 15) >> 4;
  int upPrev = bin ? 0 : NEGBIG;   // (only lane 63 ever uses the initial value)
  int b = NOMATCH;
  int c = (T - nblk * 16) - 63 + lane;
  int outv = 0;
  auto ld_chunk = [&](int blk) { const int ci = T - blk * 16 - 16 + (lane & 15); return (blk >= 0 && ci >= 0 && ci < n) ? (int)ref[ci] : NOMATCH; };
  auto ld_bnd = [&](int blk) {   // lane 63's column = ci + 1
    const int ci = T - blk * 16 - 16 + (lane & 15);
    return (bin && blk >= 0 && ci + 1 >= 0 && ci + 1 <= n) ? bin[ci + 1] : NEGBIG;
  };
  uint32_t wn[K];
#pragma unroll
  for (int i = 0; i < K; ++i) wn[i] = ld_scratch(&stack[((size_t)(nblk - 1) * K + i) * WAVE + lane]);
  int chunk = ld_chunk(nblk - 1), bchunk = ld_bnd(nblk - 1);
  for (int blk = nblk - 1; blk >= 0; --blk) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const uint32_t w = wn[i];
      const uint32_t hi = (w >> 1) & 0x55555555u, lo = w & 0x55555555u;
      dw[i] = (hi & ~lo) | ((hi & lo) << 1);
    }
    // next block's code words / letters / boundary values are in flight while this block computes
    if (blk > 0) {
#pragma unroll
      for (int i = 0; i < K; ++i) wn[i] = ld_scratch(&stack[((size_t)(blk - 1) * K + i) * WAVE + lane]);
    }
    const int chunk_n = ld_chunk(blk - 1), bchunk_n = ld_bnd(blk - 1);
#pragma unroll
    for (int f = 15; f >= 0; --f) {
      const int newc = __builtin_amdgcn_readlane(chunk, 15 - f);
      const int bnd = __builtin_amdgcn_readlane(bchunk, 15 - f);
      b = dpp_from_next(b, newc);
      const int recv = dpp_from_next(h[0], bnd);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)n) {
        const int cinv = LR_CINV - c;
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
          const int x = diag + ((a[i] == b) ? (2 << LR_CSHIFT) : 0);
          const int z = h[i] + hg[i];
          const int nv = max3i(x, up, z);
          diag = h[i];
          up = nv;
          h[i] = nv;
          bm[i] = max(bm[i], nv);
          const int delta = (int)((dw[i] >> (2 * f)) & 3u);
          g[i] = g[i] - (delta << LR_CSHIFT);
          bestkey[i] = max(bestkey[i], bm[i] + g[i] + cinv);
        }
      }
      upPrev = recv;
      if (bout) outv = writelane16(__builtin_amdgcn_readlane(h[0], 0), 15 - f, outv);
    }
    if (bout) {   // lane j holds lane 0's value of column T - 63 - 16*blk - 15 + j
      const int col = T - 63 - 16 * blk - 15 + lane;
      if (lane < 16 && col >= 0 && col <= n) bout[col] = outv;
    }
    chunk = chunk_n;
    bchunk = bchunk_n;
  }
  int hp = 0;
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (lane * K + i == pad_ls) hp = h[i];
  hpad = __shfl(hp, pad_ls / K);
  long long key = (long long)0x8000000000000000ll;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int gs = q * LRS + lane * K + i;
    if (bestkey[i] != (int)0x80000000) {
      const long long kk =
          ((long long)(bestkey[i] >> LR_CSHIFT) << 32) | ((long long)gs << LR_CSHIFT) | (long long)(bestkey[i] & LR_CINV);
      key = kk > key ? kk : key;
    }
  }
  return wave_max64(key);
}

// direction pass of strip q (natural slots: row = q*320 + ls), rows <= rmax, columns 1..ncols
__device__ __noinline__ void lr_pass_dir(const uint8_t* rowstr, const uint8_t* colstr, int m, int q, int rmax, int ncols,
                                         const int32_t* bin, int32_t* bout, uint32_t* dirs, int lane) {
  constexpr int K = LRK;
  int a[K], hg[K], h[K];
  uint32_t acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int row = q * LRS + lane * K + i;
    a[i] = (row >= 1 && row <= m) ? (int)rowstr[row - 1] : NOMATCH;
    hg[i] = (row >= 1 && row < m) ? -1 : 0;
    h[i] = 0;
    acc[i] = 0;
  }
  const int lastlane = min(WAVE - 1, (rmax - q * LRS) / K);
  const int T = ncols + lastlane;
  const int nblk = (T + 15) >> 4;
  int upPrev = bin ? 0 : NEGBIG;
  int b = NOMATCH;
  int c = -lane;
  int outv = 0;
  auto ld_chunk = [&](int blk) { const int ci = blk * 16 + (lane & 15); return (ci < ncols) ? (int)colstr[ci] : NOMATCH; };
  auto ld_bnd = [&](int blk) { const int ci = blk * 16 + (lane & 15); return (bin && ci + 1 <= ncols) ? bin[ci + 1] : NEGBIG; };
  int chunk = ld_chunk(0), bchunk = ld_bnd(0);
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_n = ld_chunk(blk + 1), bchunk_n = ld_bnd(blk + 1);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      const int bnd = __builtin_amdgcn_readlane(bchunk, f);
      b = dpp_from_prev(b, newc);
      const int recv = dpp_from_prev(h[K - 1], bnd);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)ncols) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const int x = diag + ((a[i] == b) ? 2 : 0);
          const int z = h[i] + hg[i];
          const int nv = max3i(x, up, z);
          const uint32_t code = (nv == up) ? 1u : ((nv == z) ? 2u : 0u);
          diag = h[i];
          up = nv;
          h[i] = nv;
          acc[i] |= code << (2 * f);
        }
      }
      upPrev = recv;
      if (bout) outv = writelane16(__builtin_amdgcn_readlane(h[K - 1], 63), f, outv);
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      dirs[((size_t)blk * K + i) * WAVE + lane] = acc[i];
      acc[i] = 0;
    }
    if (bout) {
      const int col = blk * 16 + lane - 62;
      if (lane < 16 && col >= 0 && col <= ncols) bout[col] = outv;
    }
    chunk = chunk_n;
    bchunk = bchunk_n;
  }
}

// unit-cost NW distance strip (rows = target letters, natural slots; E[r][0] = r, E[0][c] = c):
// the orientation test of src/split.h:564-572 (edlib NW, DISTANCE).  Returns E of local slot
// `want_ls` at the last column.
__device__ __noinline__ int lr_pass_ed(const uint8_t* tstr, int tlen, const uint8_t* qstr, int qlen, int q,
                                       const int32_t* bin, int32_t* bout, int want_ls, int lane) {
  constexpr int K = LRK;
  constexpr int POS = 1 << 28;
  int a[K], h[K], colq[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int row = q * LRS + lane * K + i;
    a[i] = (row >= 1 && row <= tlen) ? (int)tstr[row - 1] : NOMATCH;
    h[i] = row;
    colq[i] = row;
  }
  const int lastlane = min(WAVE - 1, (tlen - q * LRS) / K);
  const int T = qlen + lastlane;
  const int nblk = (T + 15) >> 4;
  int upPrev = bin ? (q * LRS - 1) : POS;   // E[row above][0] = its row index
  int b = NOMATCH;
  int c = -lane;
  int outv = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int ci = blk * 16 + (lane & 15);
    const int chunk = (ci < qlen) ? (int)qstr[ci] : NOMATCH;
    const int bchunk = (bin && ci + 1 <= qlen) ? bin[ci + 1] : POS;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      const int bnd = __builtin_amdgcn_readlane(bchunk, f);
      b = dpp_from_prev(b, newc);
      const int recv = dpp_from_prev(h[K - 1], bnd);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)qlen) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const int x = diag + ((a[i] != b) ? 1 : 0);
          const int nv = min(min(x, up + 1), h[i] + 1);
          diag = h[i];
          up = nv;
          h[i] = nv;
        }
        if (c == qlen) {
#pragma unroll
          for (int i = 0; i < K; ++i) colq[i] = h[i];
        }
      }
      upPrev = recv;
      if (bout) outv = writelane16(__builtin_amdgcn_readlane(h[K - 1], 63), f, outv);
    }
    if (bout) {
      const int col = blk * 16 + lane - 62;
      if (lane < 16 && col >= 1 && col <= qlen) bout[col] = outv;
    }
  }
  int v = 0;
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (lane * K + i == want_ls) v = colq[i];
  return __shfl(v, want_ls / K);
}

// edlibAlign(query, target, NW, DISTANCE).editDistance for |target| > 0, |query| > 0
__device__ __forceinline__ int lr_nw_distance(const uint8_t* target, int tn, const uint8_t* query, int qn, int32_t* bnd0,
                                              int32_t* bnd1, int lane) {
  const int Q = (tn + 1 + LRS - 1) / LRS;
  int d = 0;
  for (int q = 0; q < Q; ++q) {
    const int32_t* bin = (q > 0) ? ((q & 1) ? bnd0 : bnd1) : nullptr;
    int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bnd1 : bnd0) : nullptr;
    d = lr_pass_ed(target, tn, query, qn, q, bin, bout, tn - q * LRS, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  return rfl(d);
}

// code-word geometry of the strip passes
struct GeoLR {
  const uint32_t* base;
  uint64_t strip_words;
  __device__ __forceinline__ void locate(int r, int c, size_t& wi, int& t) const {
    const int qs = r / LRS, ls = r - qs * LRS;
    const int lo = ls / LRK, i = ls - lo * LRK;
    t = c + lo - 1;
    wi = (size_t)qs * strip_words + ((size_t)(t >> 4) * LRK + i) * WAVE + lo;
  }
};

// windowed run-length traceback over the per-strip direction codes; ops (0 's', 1 'v', 2 'h') to tr[] (global)
__device__ __noinline__ int lr_traceback(const uint32_t* dirs, uint64_t strip_words, int rr, int cc, uint8_t* tr, int lane,
                                         int& tailV, int& tailH) {
  GeoLR G{dirs, strip_words};
  const int tl = traceback_runs<false>(G, rr, cc, tr, lane);
  tailV = rr;
  tailH = cc;
  return tl;
}

// direction codes of rows 0..rmax x columns 1..ncols (all strips), then the traceback
__device__ __forceinline__ int lr_dir_and_trace(const uint8_t* rowstr, const uint8_t* colstr, int m, int rmax, int ncols,
                                                uint32_t* dirs, uint64_t strip_words, int32_t* bnd0, int32_t* bnd1,
                                                uint8_t* tr, int lane, int& tailV, int& tailH) {
  const int Q = rmax / LRS + 1;
  for (int q = 0; q < Q; ++q) {
    const int32_t* bin = (q > 0) ? ((q & 1) ? bnd0 : bnd1) : nullptr;
    int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bnd1 : bnd0) : nullptr;
    lr_pass_dir(rowstr, colstr, m, q, rmax, ncols, bin, bout, dirs + (size_t)q * strip_words, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int nops = lr_traceback(dirs, strip_words, rmax, ncols, tr, lane, tailV, tailH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return nops;
}

// ---- one long-read junction per wavefront ------------------------------------------------
__device__ __forceinline__ void process_lr(const SplitArgs& A, const LrArgs& R, int j, LrLds& LL, uint8_t* ws, int lane) {
  PostRef L{LL.u.post.mV, LL.u.post.mR, LL.u.post.mE, LL.u.post.cumV, LL.u.post.cumR};
  int maskw = LR_MASKW_LDS;
  auto pick_masks = [&](int m_, int n_) {   // (call once m and n are known, before the masks are built)
    const int need = (m_ + n_ + 127) / 64 + 2;
    if (need > LR_MASKW_LDS) {
      unsigned long long* g = reinterpret_cast<unsigned long long*>(ws + R.off_masks);
      L = PostRef{g, g + LR_MASKW, g + 2 * LR_MASKW, reinterpret_cast<int32_t*>(g + 3 * LR_MASKW), reinterpret_cast<int32_t*>(g + 3 * LR_MASKW) + (LR_MASKW + 1)};
      maskw = min(LR_MASKW, need);
    } else {
      maskw = need;
    }
  };
  MyersLds<MYERS_NW>& ML = LL.u.o.myers;
  const dellyhip_junction J = A.junc[j];
  const dellyhip_params& P = A.p;
  JCtx X;
  X.j = j;
  X.out = &A.res[j];
  X.ob = A.out_blob + (size_t)j * A.out_stride;
  X.ob_off = (uint64_t)j * A.out_stride;
  X.m = A.cons_len[j];
  X.n = 0;
  X.svt = J.svt;
  X.svS = J.sv_start;
  X.svE = J.sv_end;
  X.sBeg = X.sEnd = X.eBeg = X.eEnd = 0;
  X.direct = false;
  X.consLeft = X.refLeft = X.refRight = X.consRight = 0;
  StrPtr S{ws, ws + R.off_rcons, ws + R.off_ref, ws + R.off_rref};
  int32_t* bnd0 = reinterpret_cast<int32_t*>(ws + R.off_bnd0);
  int32_t* bnd1 = reinterpret_cast<int32_t*>(ws + R.off_bnd1);
  int32_t* brbuf = reinterpret_cast<int32_t*>(ws + R.off_br);
  uint8_t* trF = ws + R.off_trF;
  uint8_t* trR = ws + R.off_trR;
  uint32_t* stack = reinterpret_cast<uint32_t*>(ws + R.off_stack);
  const int m = X.m;
  const uint8_t* cons_g = A.cons_base + A.cons_off[j];
  const bool own_cons = (A.cons_base != A.out_blob) || (cons_g == X.ob);
  const int prior = X.out->status, support = X.out->sr_support;
  int status = 0;
  bool go = true, mlimit = false;
  if (prior) { status = prior; mlimit = true; go = false; }
  else if (m < 0 || m > LR_MMAX || m > R.mcap) { status = DELLYHIP_E_LIMIT; mlimit = true; go = false; }
  if (go) {
    for (int i = lane; i < m; i += WAVE) {
      const uint8_t ch = cons_g[i];
      S.cons[i] = ch;
      if (A.cons_base != A.out_blob) X.ob[i] = ch;   // (MSA modes: the consensus already lives in the slot)
    }
  }
  if (go && J.svt == 4) { status = DELLYHIP_E_LIMIT; go = false; }   // long-read splitAlign: edlib's Hirschberg regime
  if (go && !(P.reserved & 2) && m < 2 * P.minimum_flank_size + J.ins_len) go = false;     // split.h:647
  Seg seg[3];
  int nseg = 0, n = 0;
  if (go) {
    int sBeg, sEnd, eBeg, eEnd;
    if (!window_segments<false>(A, J, m, seg, nseg, sBeg, sEnd, eBeg, eEnd)) go = false;
    X.sBeg = sBeg; X.sEnd = sEnd; X.eBeg = eBeg; X.eEnd = eEnd;
    for (int q = 0; q < nseg; ++q) n += seg[q].len;
    if (go && (n > LR_NMAX || n > R.ncap)) { status = DELLYHIP_E_LIMIT; go = false; }
    if (go) {
      int o = 0;
      for (int q = 0; q < nseg; ++q) {
        fill_segment(S.ref + o, seg[q], lane);
        o += seg[q].len;
      }
    }
  }
  X.n = n;
  if (lane == 0) {
    dellyhip_result Rr;
    int* rp = reinterpret_cast<int*>(&Rr);
#pragma unroll
    for (unsigned q = 0; q < sizeof(Rr) / 4; ++q) rp[q] = 0;
    Rr.svid = J.svid;
    Rr.sv_start = J.sv_start;
    Rr.sv_end = J.sv_end;
    Rr.ins_len = J.ins_len;
    Rr.score_unsplit = Rr.score_best = Rr.cons_left = Rr.ref_left = Rr.ref_right = -1;
    Rr.matches = Rr.mismatches = -1;
    Rr.cons_len = mlimit ? 0 : m;
    Rr.cons_off = X.ob_off;
    Rr.sr_support = support;
    Rr.status = status;
    Rr.ref_len = n;
    *X.out = Rr;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  go = rfl((int)go) != 0;
#ifdef DH_LR_TIMING
  const unsigned long long tq0 = wall_clock64();
#endif
  int err_est = -1;   // consensus errors estimated from the orientation test (unknown without it)
  if (go && R.realign && m > 0 && n > 0) {
    myers_lut_init(ML.lut, lane);   // (the LDS is shared with the later phases of the previous junction)
    __syncthreads();
    // split.h:564-572: keep the orientation with the smaller NW edit distance to the window
    for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (bit-vector distance, myers_kernel.hpp; the plain strip recurrence lr_nw_distance gives the same numbers)
    // (bit-vector distance, myers_kernel.hpp; pattern = the shorter string, the distance is symmetric; beyond the rows
    //  of one pass the pattern is cut into strips whose boundary deltas park in the boundary-row arrays)
    int dF, dR;
    if (min(m, n) <= MYERS_ROWS) {
      if (m <= n) {
        // both orientations in one pass over the window (two patterns in lock-step)
        bool two;
        if (m <= WAVE * 32) two = myers_nw_fast2<1>(reinterpret_cast<MyersLds<1>&>(ML), LL.u.o.eqB, S.cons, S.rcons, m, S.ref, n, lane, dF, dR);
        else if (m <= WAVE * 64) two = myers_nw_fast2<2>(reinterpret_cast<MyersLds<2>&>(ML), LL.u.o.eqB, S.cons, S.rcons, m, S.ref, n, lane, dF, dR);
        else two = myers_nw_fast2<3>(ML, LL.u.o.eqB, S.cons, S.rcons, m, S.ref, n, lane, dF, dR);
        if (!two) {   // a consensus byte outside ACGTN: exact-compare passes
          dF = myers_nw(S.cons, m, S.ref, n, lane);
          dR = myers_nw(S.rcons, m, S.ref, n, lane);
        }
        dF = rfl(dF);
        dR = rfl(dR);
      } else {
        dF = rfl(myers_nw_auto(ML, S.ref, n, S.cons, m, lane));
        dR = rfl(myers_nw_auto(ML, S.ref, n, S.rcons, m, lane));
      }
    } else {
      int8_t* hb0 = reinterpret_cast<int8_t*>(bnd0);
      int8_t* hb1 = reinterpret_cast<int8_t*>(bnd1);
      if (m <= n) {
        dF = rfl(myers_nw_big(S.cons, m, S.ref, n, hb0, hb1, lane));
        dR = rfl(myers_nw_big(S.rcons, m, S.ref, n, hb0, hb1, lane));
      } else {
        dF = rfl(myers_nw_big(S.ref, n, S.cons, m, hb0, hb1, lane));
        dR = rfl(myers_nw_big(S.ref, n, S.rcons, m, hb0, hb1, lane));
      }
    }
    // the NW distance of the consensus to its window = the reference letters it skips (|n - m|) + its errors
    err_est = max(0, min(dF, dR) - abs(n - m));
    if (dR < dF) {   // consensus = revc
      for (int i = lane; i < m; i += WAVE) {
        const uint8_t ch = S.rcons[i];
        S.cons[i] = ch;
        if (own_cons) X.ob[i] = ch;   // (a trimmed small-inversion consensus is restored by the caller: assemble.h:850-853)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if (go) {
    for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
    for (int i = lane; i < n; i += WAVE) S.rref[i] = rc_at(S.ref, n, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  X.go = go;
  X.uniformize();

#ifdef DH_LR_TIMING
  const unsigned long long tq1 = wall_clock64();
  unsigned long long tq2 = tq1;
  int spt = 0;
#endif
  // ---- longNeedle without the dense matrices (sparse_needle.hpp) when the letters are clean and the deficit budget suffices
  bool sparse_done = false;
  int spLtot = 0, spPosC = 0;
  if (X.go && R.sparse_bytes > 0 && m >= 1 && n >= 1) {
    int dirty = 0;
    for (int i = lane; i < m; i += WAVE) dirty |= comp_acgtn(S.cons[i]) ? 0 : 1;   // (case matters: the forward pass compares raw bytes)
    for (int i = lane; i < n; i += WAVE) dirty |= comp_acgtn(S.ref[i]) ? 0 : 1;
    if (__ballot(dirty) == 0ull) {
      SparseWs W;
      W.ndp = (n + m + 2 + 63) & ~63;
      const uint64_t per_level = 2ull * W.ndp * 2 + 2ull * (uint64_t)(m + 1) * 4;
      const uint64_t runs_bytes = 4ull * 4096 * 4;
      const long long lv = (R.sparse_bytes > runs_bytes) ? (long long)((R.sparse_bytes - runs_bytes) / per_level) : 0;
      W.smax = (int)min((long long)SP_LEVELS_MAX, lv) - 1;
      // (the orientation test's NW distance says little about the consensus errors: scattered chance matches inside the
      //  skipped reference letters absorb them; the give-up rule is the prediction inside sparse_long_needle)
      (void)err_est;
      W.pred_cap = R.sparse_cost;
      uint8_t* sp = ws + R.off_sparse;
      W.runsF = reinterpret_cast<int32_t*>(sp);
      W.runsR = W.runsF + 4096;
      W.listF = W.runsR + 4096;
      W.listR = W.listF + 4096;
      W.runs_cap = 4096;
      sp += runs_bytes;
      const size_t levels = (size_t)(W.smax + 1);
      W.frF = reinterpret_cast<int16_t*>(sp);
      W.frR = W.frF + levels * W.ndp;
      W.cF = reinterpret_cast<int32_t*>(W.frR + levels * W.ndp);
      W.cR = W.cF + levels * (m + 1);
      __syncthreads();
      const SparseRes sr = sparse_long_needle<SpTile, false>(S.cons, S.rcons, S.ref, S.rref, m, n, W, LL.u.tile, LL.reachF, LL.reachR, 8, lane);
      __syncthreads();
#ifdef DH_LR_TIMING
      tq2 = wall_clock64();
      if (sr.resolved && sr.found) {   // sparse phases, units of 50 us: levels | tables | join + refRight | traces
        auto u8 = [](unsigned long long a, unsigned long long b) { return (int)min(255ull, (b - a) / 5000ull); };
        spt = u8(tq1, sr.t[0]) | (u8(sr.t[0], sr.t[1]) << 8) | (u8(sr.t[1], sr.t[3]) << 16) | (u8(sr.t[3], sr.t[4]) << 24);
      }
#endif
      if (sr.resolved) {
        sparse_done = true;
        if (lane == 0) {
          X.out->score_unsplit = sr.unsplit;
          X.out->score_best = sr.best;
          X.out->cons_left = sr.found ? sr.consLeft : 0;
          X.out->ref_left = sr.found ? sr.refLeft : 0;
          X.out->ref_right = sr.found ? sr.refRight : n;   // (no split: the last column of the free-gap row m ties its maximum)
#ifdef DH_LR_TIMING
          X.out->reserved = sr.levels;                     // diagnostic: deficit levels the sparse passes used
#endif
        }
        X.consLeft = sr.found ? sr.consLeft : 0;
        X.refLeft = sr.found ? sr.refLeft : 0;
        X.refRight = sr.found ? sr.refRight : 0;
        X.consRight = m - X.consLeft;
        X.go = sr.found != 0;
        if (sr.found) {
          const int gapref = (n - sr.refRight) - sr.refLeft;
          long long total = gapref;
          for (int i = lane; i < sr.nrunsF; i += WAVE) total += sp_ld32(W.runsF + i) & 0xffffff;
          for (int i = lane; i < sr.nrunsR; i += WAVE) total += sp_ld32(W.runsR + i) & 0xffffff;
          long long tsum = total - (lane ? gapref : 0);
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) {
            const int lo = __shfl_xor((int)(tsum & 0xffffffffll), o), hi = __shfl_xor((int)(tsum >> 32), o);
            tsum += ((long long)hi << 32) | (unsigned int)lo;
          }
          if (tsum > (long long)LR_MASKW * 64) {
            if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
            X.go = false;
          } else {
            pick_masks(m, n);
            spLtot = sparse_masks(L, W.runsF, sr.nrunsF, W.runsR, sr.nrunsR, gapref, maskw, lane, spPosC,
                                  [](PostRef& l, int pos, int cnt, unsigned long long v, unsigned long long r, int ln) { mask_append(l, pos, cnt, v, r, ln); });
            masks_finish(A, X, S, L, spLtot, spPosC, lane);
          }
        }
        X.uniformize();
      }
    }
  }
  if (sparse_done) {
#ifdef DH_LR_TIMING
    const unsigned long long tq3 = wall_clock64();
#endif
    split_detect(A, X, S, L, X.go, spLtot, spPosC, lane);
#ifdef DH_LR_TIMING
    if (lane == 0) {   // phase times in units of 10 us (wall clock 100 MHz): orientation | sparse | masks | detect
      const unsigned long long tq4 = wall_clock64();
      auto u8 = [](unsigned long long a, unsigned long long b) { return (int)min(255ull, (b - a) / 5000ull); };
      X.out->reserved = spt ? spt : (u8(tq0, tq1) | (u8(tq1, tq2) << 8) | (u8(tq2, tq3) << 16) | (u8(tq3, tq4) << 24));
    }
#endif
    return;
  }

  // ---- longNeedle: R strips, then M strips in reverse order
  const int Q = (m + 1 + LRS - 1) / LRS;
  const int pad = Q * LRS - (m + 1);
  int unsplit = 0, revmn = 0;
  long long key = (long long)0x8000000000000000ll;
  if (X.go) {
    for (int q = 0; q < Q; ++q) {
      const int32_t* bin = (q > 0) ? ((q & 1) ? bnd0 : bnd1) : nullptr;
      int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bnd1 : bnd0) : nullptr;
      const int hl = lr_pass_R(S.rcons, S.rref, m, n, q, pad, bin, bout, stack + (size_t)q * R.strip_words, brbuf, lane);
      if (q == Q - 1) revmn = hl - m;   // rev row m is the last slot of the last strip
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    for (int p = 0; p < Q; ++p) {
      const int q = Q - 1 - p;
      const int32_t* bin = (p > 0) ? ((p & 1) ? bnd0 : bnd1) : nullptr;
      int32_t* bout = (p + 1 < Q) ? ((p & 1) ? bnd1 : bnd0) : nullptr;
      int hpad = 0;
      const long long k = lr_pass_M(S.cons, S.ref, m, n, q, pad, bin, bout, stack + (size_t)q * R.strip_words, brbuf, pad,
                                    lane, hpad);
      key = k > key ? k : key;
      if (q == 0) unsplit = (hpad >> LR_CSHIFT) - m;   // M row m = rev row 0 = local slot pad of strip 0
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // ---- winner, refRight (needle.h:83-123,152)
  if (X.go) {
    const int khi = rfl((int)(key >> 32)), klo = rfl((int)(key & 0xffffffffll));
    int best = khi - m;
    const int gstar = (int)(((unsigned)klo >> LR_CSHIFT) & 0xffffu);
    int refLeft = LR_CINV - (klo & LR_CINV);
    unsplit = rfl(unsplit);
    revmn = rfl(revmn);
    int consRight = gstar - pad, consLeft = m - consRight;
    int refRight = 0;
    bool found = false;
    if (unsplit == revmn) {
      if (best <= unsplit) {
        best = unsplit;
        consLeft = 0;
        refLeft = 0;
        consRight = m;
      }
      {
        const int F = 16;
        const int gs = consRight + pad;
        const int qs = gs / LRS, lsl = gs - qs * LRS;
        const int ls = lsl / LRK, is = lsl - ls * LRK;
        const uint32_t* srow = stack + (size_t)qs * R.strip_words;
        const int Xc = n - refLeft;
        const int t = Xc + ls - 1;
        const int wtop = (t >= 0) ? t / F : -1;
        const int rounds = (Xc >= 1) ? (Xc + F * WAVE - 1) / (F * WAVE) + 1 : 0;
        int bestcol = 0;
        for (int r = 0; r < rounds; ++r) {
          const int widx = wtop - (r * WAVE + lane);
          int cand = 0;
          if (widx >= 0) {
            const uint32_t w = ld_scratch(&srow[((size_t)widx * LRK + is) * WAVE + ls]);
            const int fmax = min(F - 1, t - widx * F);
            const int fmin = max(0, ls - widx * F);
            if (fmax >= fmin) {
              uint32_t keep = (2 * fmax + 2 >= 32) ? 0xffffffffu : ((1u << (2 * fmax + 2)) - 1u);
              keep &= ~((1u << (2 * fmin)) - 1u);
              const uint32_t x = w & keep;
              if (x) cand = widx * F + ((31 - __builtin_clz(x)) >> 1) - ls + 1;
            }
          }
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) cand = max(cand, __shfl_xor(cand, o));
          bestcol = max(bestcol, cand);
        }
        refRight = bestcol;
      }
      found = (best != unsplit);
      if (lane == 0) {
        X.out->score_best = best;
        X.out->cons_left = consLeft;
        X.out->ref_left = refLeft;
        X.out->ref_right = refRight;
      }
    }
    if (lane == 0) X.out->score_unsplit = unsplit;
    X.consLeft = consLeft;
    X.refLeft = refLeft;
    X.refRight = refRight;
    X.consRight = consRight;
    X.go = found;
    X.uniformize();
  }
  // ---- tracebacks on recomputed direction codes, column masks, split detection
  go = X.go;
  int Ltot = 0, posC = 0;
  if (go) {
    const int consLeft = X.consLeft, refLeft = X.refLeft, consRight = X.consRight, refRight = X.refRight;
    int nF = 0, tvF = 0, thF = 0, nR = 0, tvR = 0, thR = 0;
    if (consLeft > 0 && refLeft > 0)
      nF = lr_dir_and_trace(S.cons, S.ref, m, consLeft, refLeft, stack, R.strip_words, bnd0, bnd1, trF, lane, tvF, thF);
    else { tvF = consLeft; thF = (consLeft > 0) ? 0 : refLeft; }
    if (consRight > 0 && refRight > 0)
      nR = lr_dir_and_trace(S.rcons, S.rref, m, consRight, refRight, stack, R.strip_words, bnd0, bnd1, trR, lane, tvR, thR);
    else { tvR = consRight; thR = (consRight > 0) ? 0 : refRight; }
    const int gapref = (n - refRight) - refLeft;
    const long long total = (long long)thF + tvF + nF + gapref + nR + tvR + thR;
    if (total > (long long)LR_MASKW * 64) {
      if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
      go = false;
    } else {
      pick_masks(m, n);
      Ltot = needle_masks(L, trF, nF, tvF, thF, trR, nR, tvR, thR, gapref, maskw, lane, posC);
      masks_finish(A, X, S, L, Ltot, posC, lane);
    }
  }
  split_detect(A, X, S, L, go, Ltot, posC, lane);
}

__global__ __launch_bounds__(WAVE) void lr_kernel(SplitArgs A, LrArgs R) {
  __shared__ LrLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = R.ws + (size_t)blockIdx.x * R.ws_stride;
  // junction latencies differ by an order of magnitude (levels of the sparse passes, dense fallback): the wavefronts
  // pull from the host-sorted list (largest consensus x window first) instead of striding over it
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    const int j = A.work_list[w];
    if (j >= 0) process_lr(A, R, j, L, ws, lane);
    __syncthreads();
  }
}

}  // namespace dh
