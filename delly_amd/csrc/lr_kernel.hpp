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
  // dense strips on a TEAM of wavefronts (lr_dense_team_kernel, running beside lr_kernel on a second stream): lr_kernel hands the
  // junctions its sparse passes give up on to the list in team_state instead of sweeping the strips on its one wavefront
  int32_t* team_state;    // nullptr: no team kernel.  [LRT_COUNT] junctions listed, [LRT_TAKEN] claimed by teams,
                          // [LRT_ERROR]; the list (junction indices, -1 = not yet written) from [LRT_LIST]
  uint64_t off_bndx;      // LR_TEAM_W - 1 more boundary rows behind bnd0 / bnd1 (a team keeps LR_TEAM_W + 1 strips' rows alive)
  int32_t team_first_ws;  // workspace index of team 0 (the teams' workspaces follow lr_kernel's)
  int32_t team_cap;       // entries of the list; a junction that finds it full stays on lr_kernel's wavefront
  int32_t lr_grid;        // wavefronts of lr_kernel: every one of them ends with ONE fetch beyond the work list, so the work counter
                          // reads n_work + lr_grid exactly when lr_kernel is through with every junction
};
#ifndef DH_LR_TEAM_W
#define DH_LR_TEAM_W 4
#endif
constexpr int LR_TEAM_W = DH_LR_TEAM_W;   // wavefronts of a team (2 / 6 / 8 measured: CHANGELOG.md 3.7)
enum { LRT_COUNT = 0, LRT_TAKEN = 1, LRT_SPARE = 2, LRT_ERROR = 3, LRT_LIST = 4 };
#ifdef DH_LR_TEAM_DEBUG
constexpr int LRT_DBG_INTS = 8 * 4096 + 8;   // time-line marks behind the list (8 ints per team) + lr_kernel's start
#else
constexpr int LRT_DBG_INTS = 0;
#endif

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

// Strips of one matrix on DIFFERENT wavefronts of a team: strip q + 1 needs the last row of strip q, column by column, so it
// can run a few blocks of 16 columns behind it.  Each strip counts its finished blocks in LDS; the boundary rows go through
// the workspace as before (same CU, same L1: the counter is a workgroup-scope release / acquire).  A block of the producer
// ends 62 columns behind the consumer's first lane, so the values of block b are there once the producer has finished b + 5.
struct LrPipe {
  int* prev = nullptr;      // finished blocks of the strip whose last row is read (nullptr: none, or the strips run one after the other)
  int* mine = nullptr;      // this strip's counter (nullptr: nobody reads its last row)
  int prev_nblk = 0;        // blocks the producer runs in total
  int* bail = nullptr;      // set when a wait ran out of patience: a team member is gone, everybody stops waiting
};
#ifndef DH_LR_TEAM_PATIENCE
#define DH_LR_TEAM_PATIENCE 400000000ull   // 4 s of the 100 MHz wall clock
#endif
constexpr unsigned long long LR_TEAM_PATIENCE = DH_LR_TEAM_PATIENCE;
__device__ __forceinline__ void lr_pipe_wait(const LrPipe& P, int blk) {   // before the boundary values of block `blk` are loaded
  if (!P.prev) return;
  const int need = min(P.prev_nblk, blk + 5);
  volatile int* pv = P.prev;
  if (*pv < need) {
    const unsigned long long t0 = wall_clock64();
    while (*pv < need) {
      __builtin_amdgcn_s_sleep(2);
      if (*(volatile int*)P.bail) break;
      if (wall_clock64() - t0 > LR_TEAM_PATIENCE) { *(volatile int*)P.bail = 1; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lr_pipe_post(const LrPipe& P, int blocks_done) {
  if (!P.mine) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the boundary stores of the block first)
  *(volatile int*)P.mine = blocks_done;
}

// R pass of strip q (rev rows rho = q*320 + ls - pad).  bin/bout: boundary rows (V' of the row
// above the strip / of the strip's last row), index = column.  Pushes the running-max codes of
// pass_R (split_kernel.hpp) to `stack`, the final running maxima to brout[q*320 + ls].
// Returns the final V' of the strip's last slot (lane 63).
__device__ __noinline__ int lr_pass_R(const uint8_t* rcons, const uint8_t* rref, int m, int n, int q, int pad,
                                      const int32_t* bin, int32_t* bout, uint32_t* stack, int32_t* brout, int lane, LrPipe P = LrPipe{}) {
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
  lr_pipe_wait(P, 0);
  int chunk = ld_chunk(0), bchunk = ld_bnd(0);   // column of lane 0 at step 16*blk+f is 16*blk+f+1
  for (int blk = 0; blk < nblk; ++blk) {
    lr_pipe_wait(P, blk + 1);
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
    lr_pipe_post(P, blk + 1);
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
                                            const int32_t* brin, int pad_ls, int lane, int& hpad, LrPipe P = LrPipe{}) {
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
  lr_pipe_wait(P, 0);   // (the blocks run from nblk - 1 down: block blk is number nblk - 1 - blk of this pass)
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
    lr_pipe_wait(P, nblk - blk);
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
    lr_pipe_post(P, nblk - blk);
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
                                         const int32_t* bin, int32_t* bout, uint32_t* dirs, int lane, LrPipe P = LrPipe{}) {
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
  lr_pipe_wait(P, 0);
  int chunk = ld_chunk(0), bchunk = ld_bnd(0);
  for (int blk = 0; blk < nblk; ++blk) {
    lr_pipe_wait(P, blk + 1);
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
    lr_pipe_post(P, blk + 1);
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
    DH_SYNC();
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

// -DDH_LR_TEAM_DEBUG (tools/lr_team_check.py): the time line of every team behind the list, 8 ints per team, in units of 10 us of the
// 100 MHz wall clock: [0] end, [1] junction, [2] the workgroup started, [3] junction obtained, [4] set-up done, [5] R strips, [6] M strips,
// [7] winner; lr_kernel's block 0 leaves its own start behind the marks.  dellyhip_batch_lr_team_stats prints them.
#ifdef DH_LR_TEAM_DEBUG
#define LRT_MARK(code, val) do { if (lane == 0) { int32_t* dbg_ = R.team_state + LRT_LIST + R.team_cap + 8 * (int)blockIdx.x; const int now_ = (int)((wall_clock64() / 1000ull) & 0x7fffffffull); if ((code) == 1) dbg_[1] = (val); if ((code) >= 1 && (code) <= 5) dbg_[2 + (code)] = now_; if ((code) == 9) dbg_[0] = now_; if ((code) == 0) dbg_[2] = now_; } } while (0)
#else
#define LRT_MARK(code, val) do { } while (0)
#endif
// ---- a team of wavefronts for the dense strips -------------------------------------------------------------------------
// lr_dense_team_kernel: LR_TEAM_W wavefronts per workgroup.  Wavefront 0 ("main") runs the junction as lr_kernel would; the
// others only run strips: main writes a command (which matrix, its strings and buffers) into LDS, every wavefront w takes the
// strips w, w + W, ... of it, and consecutive strips run pipelined (LrPipe) instead of one after the other.  A team keeps
// W + 1 boundary rows: strip q writes row q % (W + 1), which strip q + W + 1 -- the same wavefront as its reader q + 1,
// later -- overwrites.
enum { LRC_EXIT = 0, LRC_R = 1, LRC_M = 2, LRC_DIR = 3 };
struct LrTeamCtl {
  int seq, cmd, done, bail;
  int prog[LR_QMAX + 2];             // finished blocks per strip of the running command
  const uint8_t* rowstr;
  const uint8_t* colstr;
  int m, n, Q, pad, rmax, pad_ls;    // (n = columns of the command)
  uint32_t* codes;                   // R / M: running-max codes, DIR: direction codes; strip q at q * strip_words
  uint64_t strip_words;
  int32_t* br;
  int32_t* bnd[LR_TEAM_W + 1];
  long long key[LR_TEAM_W];          // M: best join key of a wavefront's strips
  int hl, hpad;                      // R: final V' of the last strip; M: final V' of M row m
};

__device__ __forceinline__ void lr_team_strips(LrTeamCtl& C, int wv, int lane) {
  constexpr int W = LR_TEAM_W;
  const int cmd = rfl(C.cmd), Q = rfl(C.Q), m = rfl(C.m), n = rfl(C.n), pad = rfl(C.pad);
  const uint8_t* rowstr = C.rowstr;
  const uint8_t* colstr = C.colstr;
  uint32_t* codes = C.codes;
  const uint64_t strip_words = C.strip_words;
  long long key = (long long)0x8000000000000000ll;
  for (int p = wv; p < Q; p += W) {
    const int32_t* bin = (p > 0) ? C.bnd[(p - 1) % (W + 1)] : nullptr;
    int32_t* bout = (p + 1 < Q) ? C.bnd[p % (W + 1)] : nullptr;
    LrPipe P;
    P.prev = (p > 0) ? &C.prog[p - 1] : nullptr;
    P.mine = (p + 1 < Q) ? &C.prog[p] : nullptr;
    P.prev_nblk = (n + 63 + 15) >> 4;
    P.bail = &C.bail;
    if (cmd == LRC_R) {
      const int hl = lr_pass_R(rowstr, colstr, m, n, p, pad, bin, bout, codes + (size_t)p * strip_words, C.br, lane, P);
      if (p == Q - 1 && lane == 0) C.hl = hl;
    } else if (cmd == LRC_M) {
      const int q = Q - 1 - p;
      int hpad = 0;
      const long long k = lr_pass_M(rowstr, colstr, m, n, q, pad, bin, bout, codes + (size_t)q * strip_words, C.br, rfl(C.pad_ls), lane, hpad, P);
      key = k > key ? k : key;
      if (q == 0 && lane == 0) C.hpad = hpad;
    } else {
      lr_pass_dir(rowstr, colstr, m, p, rfl(C.rmax), n, bin, bout, codes + (size_t)p * strip_words, lane, P);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DH_SYNC();
  }
  if (cmd == LRC_M && lane == 0) C.key[wv] = key;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  DH_SYNC();
}

// main wavefront: hand the command in C to the helpers, run the own share of its strips, wait for theirs
__device__ __forceinline__ void lr_team_run(LrTeamCtl& C, int cmd, int lane) {
  if (lane < LR_QMAX + 2) C.prog[lane] = 0;
  if (lane < LR_TEAM_W) C.key[lane] = (long long)0x8000000000000000ll;
  if (lane == 0) { C.cmd = cmd; C.done = 0; }
  DH_SYNC();
  if (lane == 0) *(volatile int*)&C.seq = C.seq + 1;
  lr_team_strips(C, 0, lane);
  volatile int* dn = &C.done;
  if (*dn < LR_TEAM_W - 1) {
    const unsigned long long t0 = wall_clock64();
    while (*dn < LR_TEAM_W - 1) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > LR_TEAM_PATIENCE) { *(volatile int*)&C.bail = 1; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the other wavefronts of a team: strips on command until main says LRC_EXIT
__device__ __forceinline__ void lr_team_helper(LrTeamCtl& C, int wv, int lane) {
  int seen = 0;
  for (;;) {
    volatile int* sq = &C.seq;
    {   // (main always ends with LRC_EXIT; it waits for lr_kernel, which does not wait for anybody -- the limit is for a dead main only)
      const unsigned long long t0 = wall_clock64();
      while (*sq == seen) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200ull * LR_TEAM_PATIENCE) return;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    seen = rfl(*sq);
    if (rfl(C.cmd) == LRC_EXIT) return;
    lr_team_strips(C, wv, lane);
    if (lane == 0) __hip_atomic_fetch_add(&C.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// direction codes of rows 0..rmax x columns 1..ncols (all strips), then the traceback
__device__ __forceinline__ int lr_dir_and_trace(const uint8_t* rowstr, const uint8_t* colstr, int m, int rmax, int ncols,
                                                uint32_t* dirs, uint64_t strip_words, int32_t* bnd0, int32_t* bnd1,
                                                uint8_t* tr, int lane, int& tailV, int& tailH, LrTeamCtl* C = nullptr) {
  const int Q = rmax / LRS + 1;
  if (C && Q > 1) {   // the strips on the team
    if (lane == 0) {
      C->rowstr = rowstr; C->colstr = colstr; C->m = m; C->n = ncols; C->Q = Q; C->pad = 0; C->rmax = rmax; C->pad_ls = 0;
      C->codes = dirs; C->strip_words = strip_words;
    }
    lr_team_run(*C, LRC_DIR, lane);
  } else {
    for (int q = 0; q < Q; ++q) {
      const int32_t* bin = (q > 0) ? ((q & 1) ? bnd0 : bnd1) : nullptr;
      int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bnd1 : bnd0) : nullptr;
      lr_pass_dir(rowstr, colstr, m, q, rmax, ncols, bin, bout, dirs + (size_t)q * strip_words, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DH_SYNC();
    }
  }
  const int nops = lr_traceback(dirs, strip_words, rmax, ncols, tr, lane, tailV, tailH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DH_SYNC();
  return nops;
}

// ---- one long-read junction per wavefront ------------------------------------------------
// TEAM false: lr_kernel (one wavefront; with R.team_state set, a junction the sparse passes give up on goes to the teams' list);
// TEAM true: the main wavefront of a team of lr_dense_team_kernel (no sparse attempt, the strips through C)
template <bool TEAM = false>
__device__ __forceinline__ void process_lr(const SplitArgs& A, const LrArgs& R, int j, LrLds& LL, uint8_t* ws, int lane, LrTeamCtl* C = nullptr) {
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
  X.direct = (A.ref_base != nullptr);   // dellyhip_long_needle beyond the short-read shapes: s2 given, longNeedle only (round 6)
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
  if (go && !X.direct && !(P.reserved & 2) && m < 2 * P.minimum_flank_size + J.ins_len) go = false;     // split.h:647
  Seg seg[3];
  int nseg = 0, n = 0;
  if (go && X.direct) {
    n = A.ref_len[j];
    if (n < 0 || n > LR_NMAX || n > R.ncap) { status = DELLYHIP_E_LIMIT; go = false; n = 0; }
    else {
      const uint8_t* rg = A.ref_base + A.ref_off[j];
      for (int i = lane; i < n; i += WAVE) S.ref[i] = rg[i];
    }
  } else if (go) {
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
  DH_SYNC();
  go = rfl((int)go) != 0;
#ifdef DH_LR_TIMING
  const unsigned long long tq0 = wall_clock64();
#endif
  int err_est = -1;   // consensus errors estimated from the orientation test (unknown without it)
  if (go && R.realign && m > 0 && n > 0) {
    myers_lut_init(ML.lut, lane);   // (the LDS is shared with the later phases of the previous junction)
    DH_SYNC();
    // split.h:564-572: keep the orientation with the smaller NW edit distance to the window
    for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DH_SYNC();
    // (bit-vector distance, myers_kernel.hpp; the plain strip recurrence lr_nw_distance gives the same numbers)
    // (bit-vector distance, myers_kernel.hpp; pattern = the shorter string, the distance is symmetric; beyond the rows
    //  of one pass the pattern is cut into strips whose boundary deltas park in the boundary-row arrays)
    int dF, dR;
#ifdef DH_LR_TIMING
    const unsigned long long tmy0 = wall_clock64();
#endif
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
#ifdef DH_LR_TIMING
    if (lane == 0 && (j & 255) == 0) printf("lr junction %d: m %d n %d, orientation passes %llu us (since the junction's start %llu us)\n", j, m, n, (wall_clock64() - tmy0) / 100, (wall_clock64() - tq0) / 100);
#endif
    // the NW distance of the consensus to its window = the reference letters it skips (|n - m|) + its errors
    err_est = max(0, min(dF, dR) - abs(n - m));
    if (dR < dF) {   // consensus = revc
      for (int i = lane; i < m; i += WAVE) {
        const uint8_t ch = S.rcons[i];
        S.cons[i] = ch;
        if (own_cons) X.ob[i] = ch;   // (a trimmed small-inversion consensus is restored by the caller: assemble.h:850-853)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DH_SYNC();
    }
  }
  if (go) {
    for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
    for (int i = lane; i < n; i += WAVE) S.rref[i] = rc_at(S.ref, n, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DH_SYNC();
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
  if (!TEAM && X.go && R.sparse_bytes > 0 && m >= 1 && n >= 1) {
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
      DH_SYNC();
      const SparseRes sr = sparse_long_needle<SpTile, false>(S.cons, S.rcons, S.ref, S.rref, m, n, W, LL.u.tile, LL.reachF, LL.reachR, 8, lane);
      DH_SYNC();
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
    if (X.go && X.direct && lane == 0) X.out->ok = 1;   // longNeedle() returned true, rows written by masks_finish
    split_detect(A, X, S, L, X.go && !X.direct, spLtot, spPosC, lane);
#ifdef DH_LR_TIMING
    if (lane == 0) {   // phase times in units of 10 us (wall clock 100 MHz): orientation | sparse | masks | detect
      const unsigned long long tq4 = wall_clock64();
      auto u8 = [](unsigned long long a, unsigned long long b) { return (int)min(255ull, (b - a) / 5000ull); };
      X.out->reserved = spt ? spt : (u8(tq0, tq1) | (u8(tq1, tq2) << 8) | (u8(tq2, tq3) << 16) | (u8(tq3, tq4) << 24));
    }
#endif
    return;
  }

  // The dense strips of this junction on a team (lr_dense_team_kernel)?  The junction then runs through the rest of this function
  // with go = false -- a record that says "not refined", as for any junction without a window -- and its index is put on the
  // teams' list at the very end: whatever this wavefront writes is written before the team starts and is written again there.
  int defer_slot = -1;
  if (!TEAM && R.team_state != nullptr && X.go) {
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&R.team_state[LRT_COUNT], 1);
    slot = rfl(slot);
    if (slot < R.team_cap) {   // (a full list: this wavefront sweeps the strips itself, as without teams)
      defer_slot = slot;
      X.go = false;
      X.uniformize();
    }
  }
  // ---- longNeedle: R strips, then M strips in reverse order
  const int Q = (m + 1 + LRS - 1) / LRS;
  const int pad = Q * LRS - (m + 1);
  int unsplit = 0, revmn = 0;
  long long key = (long long)0x8000000000000000ll;
#ifdef DH_LR_TIMING
  const unsigned long long td0 = wall_clock64();
  unsigned long long td1 = td0, td2 = td0, td3 = td0;
#endif
  if (TEAM) LRT_MARK(2, X.go);
  if (TEAM && X.go) {
    if (lane == 0) {
      C->rowstr = S.rcons; C->colstr = S.rref; C->m = m; C->n = n; C->Q = Q; C->pad = pad; C->rmax = 0; C->pad_ls = pad;
      C->codes = stack; C->strip_words = R.strip_words; C->br = brbuf;
    }
    lr_team_run(*C, LRC_R, lane);
    if (TEAM) LRT_MARK(3, Q);
    revmn = rfl(C->hl) - m;
    DH_SYNC();
    if (lane == 0) { C->rowstr = S.cons; C->colstr = S.ref; }
    lr_team_run(*C, LRC_M, lane);
    for (int w = 0; w < LR_TEAM_W; ++w) {
      const long long k = C->key[w];
      key = k > key ? k : key;
    }
    unsplit = (rfl(C->hpad) >> LR_CSHIFT) - m;
    DH_SYNC();
    if (TEAM) LRT_MARK(4, unsplit);
  } else if (X.go) {
    for (int q = 0; q < Q; ++q) {
      const int32_t* bin = (q > 0) ? ((q & 1) ? bnd0 : bnd1) : nullptr;
      int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bnd1 : bnd0) : nullptr;
      const int hl = lr_pass_R(S.rcons, S.rref, m, n, q, pad, bin, bout, stack + (size_t)q * R.strip_words, brbuf, lane);
      if (q == Q - 1) revmn = hl - m;   // rev row m is the last slot of the last strip
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DH_SYNC();
    }
#ifdef DH_LR_TIMING
    td1 = wall_clock64();
#endif
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
      DH_SYNC();
    }
  }
#ifdef DH_LR_TIMING
  td2 = wall_clock64();
#endif
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
  if (TEAM) LRT_MARK(5, X.go);
  // ---- tracebacks on recomputed direction codes, column masks, split detection
  go = X.go;
  int Ltot = 0, posC = 0;
  if (go) {
    const int consLeft = X.consLeft, refLeft = X.refLeft, consRight = X.consRight, refRight = X.refRight;
    int nF = 0, tvF = 0, thF = 0, nR = 0, tvR = 0, thR = 0;
    if (consLeft > 0 && refLeft > 0)
      nF = lr_dir_and_trace(S.cons, S.ref, m, consLeft, refLeft, stack, R.strip_words, bnd0, bnd1, trF, lane, tvF, thF, TEAM ? C : nullptr);
    else { tvF = consLeft; thF = (consLeft > 0) ? 0 : refLeft; }
    if (consRight > 0 && refRight > 0)
      nR = lr_dir_and_trace(S.rcons, S.rref, m, consRight, refRight, stack, R.strip_words, bnd0, bnd1, trR, lane, tvR, thR, TEAM ? C : nullptr);
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
#ifdef DH_LR_TIMING
  td3 = wall_clock64();
#endif
  if (go && X.direct && lane == 0) X.out->ok = 1;   // longNeedle() returned true
  split_detect(A, X, S, L, go && !X.direct, Ltot, posC, lane);
  if (!TEAM && defer_slot >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence();
    if (lane == 0) atomicExch(&R.team_state[LRT_LIST + defer_slot], j);
  }
#ifdef DH_LR_TIMING
  if (lane == 0) {   // dense path, units of 200 us: set-up + sparse attempt | R strips | M strips | winner + direction passes + traces + masks
    auto u8 = [](unsigned long long a, unsigned long long b) { return (int)min(255ull, (b - a) / 20000ull); };
    X.out->reserved = u8(tq0, td0) | (u8(td0, td1) << 8) | (u8(td1, td2) << 16) | (u8(td2, td3) << 24);
  }
#endif
}

__global__ __launch_bounds__(WAVE) void lr_kernel(SplitArgs A, LrArgs R) {
  __shared__ LrLds L;
  const int lane = threadIdx.x;
#ifdef DH_LR_TEAM_DEBUG
  if (R.team_state && blockIdx.x == 0 && lane == 0) R.team_state[LRT_LIST + R.team_cap + 8 * 4096] = (int)((wall_clock64() / 1000ull) & 0x7fffffffull);
#endif
  uint8_t* ws = R.ws + (size_t)blockIdx.x * R.ws_stride;
  // junction latencies differ by an order of magnitude (levels of the sparse passes, dense fallback): the wavefronts
  // pull from the host-sorted list (largest consensus x window first) instead of striding over it
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    const int j = A.work_list[w];
#if defined(DH_LR_TIMING) && DH_LR_TIMING == 2   // when a junction starts and ends, units of 50 us of the 100 MHz wall clock (low 16 bits each)
    const unsigned long long tj0 = wall_clock64();
#endif
    if (j >= 0) process_lr(A, R, j, L, ws, lane);
#if defined(DH_LR_TIMING) && DH_LR_TIMING == 2
    if (j >= 0 && lane == 0) A.res[j].reserved = (int)((((tj0 / 5000ull) & 0xffffull) << 16) | ((wall_clock64() / 5000ull) & 0xffffull));
#endif
    DH_SYNC();
  }
}

// The dense strips of the junctions lr_kernel listed, one junction per team of LR_TEAM_W wavefronts.  Launched on a second
// stream right after lr_kernel so that the teams work while lr_kernel is still busy with the sparse passes of the other
// junctions; a team that finds the list empty waits until lr_kernel has been through every junction (its work counter says so).
// If the two streams happen to run one after the other the teams simply find the complete list.
__global__ __launch_bounds__(WAVE * LR_TEAM_W) void lr_dense_team_kernel(SplitArgs A, LrArgs R) {
  __shared__ LrLds L;
  __shared__ LrTeamCtl C;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = rfl((int)(threadIdx.x / WAVE));
  if (threadIdx.x == 0) { C.seq = 0; C.cmd = LRC_EXIT; C.done = 0; C.bail = 0; }
  __syncthreads();   // (the only real barrier of this kernel)
  // the teams are the critical path of a batch (a few junctions, ~6 M instructions each) and share their SIMDs with lr_kernel's
  // wavefronts: they issue first
  __builtin_amdgcn_s_setprio(3);
  LRT_MARK(0, 0);
  if (wv != 0) {
    lr_team_helper(C, wv, lane);
    return;
  }
  uint8_t* ws = R.ws + (size_t)(R.team_first_ws + (int)blockIdx.x) * R.ws_stride;
  if (lane == 0) {
    const uint64_t row = ((uint64_t)R.ncap + 128) * 4;
    C.bnd[0] = reinterpret_cast<int32_t*>(ws + R.off_bnd0);
    C.bnd[1] = reinterpret_cast<int32_t*>(ws + R.off_bnd1);
    for (int k = 2; k <= LR_TEAM_W; ++k) C.bnd[k] = reinterpret_cast<int32_t*>(ws + R.off_bndx + (uint64_t)(k - 2) * row);
  }
  DH_SYNC();
  int32_t* ts = R.team_state;
  const int n_work = A.n_work;
  for (;;) {
    int k = 0;
    if (lane == 0) k = atomicAdd(&ts[LRT_TAKEN], 1);
    k = rfl(k);
    bool have = false;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      if (k >= R.team_cap) break;   // (beyond the list: lr_kernel keeps those junctions)
      if (__hip_atomic_load(&ts[LRT_COUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > k) { have = true; break; }
      const int dn = __hip_atomic_load(A.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dn >= n_work + R.lr_grid) {   // every wavefront of lr_kernel has made its last fetch: the count is final
        __threadfence();
        have = __hip_atomic_load(&ts[LRT_COUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > k;
        break;
      }
      // (no fence while polling: a device-scope fence writes the L2 back and invalidates it, and 64 idle teams doing that every
      //  microsecond cost lr_kernel's table loads a third of their speed)
      __builtin_amdgcn_s_sleep(127);
      if (wall_clock64() - t0 > 150ull * LR_TEAM_PATIENCE) {   // (ten minutes: lr_kernel is gone)
        if (lane == 0) atomicExch(&ts[LRT_ERROR], 1);
        break;
      }
    }
    if (!have) break;
    int j = -1;
    while ((j = __hip_atomic_load(&ts[LRT_LIST + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {   // (written right after the count)
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 160ull * LR_TEAM_PATIENCE) break;
    }
    __threadfence();
    j = rfl(j);
    if (j < 0) {
      if (lane == 0) atomicExch(&ts[LRT_ERROR], 1);
      break;
    }
    LRT_MARK(1, j);
    process_lr<true>(A, R, j, L, ws, lane, &C);
    DH_SYNC();
    LRT_MARK(9, j);
    if (rfl(*(volatile int*)&C.bail)) {   // a wait inside the team ran out: the junction's record is not to be trusted
      if (lane == 0) {
        A.res[j].status = DELLYHIP_E_RUNTIME;
        atomicExch(&ts[LRT_ERROR], 1);
      }
      break;   // (the team ends here: its command pipe is in an unknown state.  The flag fails the whole batch on the host,
               //  dellyhip_batch_sync, so the junctions still on the list are not handed out as "not refined")
    }
  }
  if (lane == 0) { C.cmd = LRC_EXIT; }
  DH_SYNC();
  if (lane == 0) *(volatile int*)&C.seq = C.seq + 1;
}

}  // namespace dh
