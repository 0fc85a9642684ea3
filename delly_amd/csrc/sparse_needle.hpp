// sparse_needle.hpp -- gfx950 device code: longNeedle (src/needle.h:45-222) WITHOUT its dense matrices.
//
// The reference fills two (m+1) x (n+1) score matrices (forward, reverse complement), their row prefix maxima, scans
// every cell for the best join and walks two tracebacks.  Everything it derives -- bestScore, (consLeft, refLeft),
// refRight, the two paths -- only involves cells whose DEFICIT D[r][c] = r - mat[r][c] (0 on a perfect prefix match;
// +2 per mismatch or consensus-only move, +1 per reference-only move) is at most s = m - bestScore:
//   * a split path through forward cell (r, c) scores at most mat[r][c] + (m - r), so cells with D > s cannot carry it;
//   * cells the tracebacks compare against SUCCESSFULLY lie on optimal paths too (a failed comparison stays failed when
//     the neighbour's value is only known to be "worse than s");
//   * the join winner, its first-row-major tie-break and refRight are decided among cells with D_F + D_R = s.
// Cells of bounded deficit are described completely by furthest-reaching tables (Landau-Vishkin / WFA style):
//   FR[d][k] = last row r on diagonal k = c - r whose deficit is <= d   (D is non-decreasing along a diagonal),
// computed level by level from FR[d-1][k-1] (reference-only move), FR[d-2][k] + 1 (mismatch), FR[d-2][k+1] + 1
// (consensus-only move), the free first row (D[0][c] = 0) and the first column (D[r][0] = 2r), each followed by a match
// extension.  Work: (s+1) x (n+m+1) table entries per matrix instead of (m+1) x (n+1) cells -- ~25x fewer at the
// long-read shapes (2 kb x 7 kb, s ~ 60), and the tracebacks need no direction matrices at all.
// Row m (free trailing gap in the reference) is handled as the reference's prefix maximum: mat[m][c] = max over c' <= c of
// the interior-rule value, so "first column with deficit <= d" is the same query as in every other row.
//
// Exactness: the CPU prototype tools/proto/sparse_needle.py restates this procedure and is bit-compared with a dense
// restatement of the reference (itself checked against oracle/) on thousands of random and adversarial inputs (repeats,
// low complexity, no-split, junk); the kernel is bit-compared with oracle/_ref by the -m gpu tests.  The procedure needs
// clean letters (A, C, G, T, N, upper case: then reverseComplement is an involution and mat[m][n] == rev[m][n] always);
// other inputs, and junctions not resolved at the deficit budget the workspace allows, take the dense strip passes.
#pragma once
#include <type_traits>

// the helpers of the sparse procedure are folded into the kernels (-DDH_SP_CALLS: real functions -- measured 6 % slower, their
// frames live in scratch memory)
#ifndef DH_SP_CALLS
#define DH_SP_FN __forceinline__
#else
#define DH_SP_FN __noinline__
#endif

#include "split_kernel.hpp"

namespace dh {

constexpr int SP_NEG = -30000;                 // "no cell of this cost on this diagonal" (int16 tables)
constexpr int SP_INF = 0x3fffffff;
constexpr int SP_LEVELS_MAX = 512;             // deficit levels 0 .. 511
constexpr int SP_UNKNOWN = -(1 << 30);         // score_unsplit when mat[m][n] lies below the deficit budget (not needed for the result)

struct SparseWs {
  int16_t* frF;      // [level][ndp]
  int16_t* frR;
  int32_t* cF;       // [level][m + 1]: first column whose prefix-min deficit is <= level (SP_INF: none)
  int32_t* cR;
  int32_t* runsF;    // traceback runs, push order: (op << 24) | length, op 0 's', 1 'v', 2 'h'
  int32_t* runsR;
  int32_t* listF;    // diagonals deep enough to meet the other side (runs_cap entries each)
  int32_t* listR;
  int32_t ndp;       // level stride of frF / frR (>= n + m + 2)
  int32_t smax;      // levels the workspace holds minus one
  int32_t pred_cap;  // give up (dense passes) when the deficit predicted after the first level block exceeds this
  int32_t runs_cap;
};

struct SparseRes {
  int resolved;      // 0: deficit budget exhausted -> dense passes
  int found;
  int unsplit;       // mat[m][n] or SP_UNKNOWN
  int best, consLeft, refLeft, refRight;
  int nrunsF, nrunsR;
  int levels;        // deficit levels used (diagnostic)
  int mmF, mmR;      // mismatch columns on the two traced paths (from the deficits: a path of deficit D with V consensus-only
                     // and H paid reference-only moves crosses (D - 2 V - H) / 2 mismatches)
  unsigned long long t[5];   // DH_LR_TIMING: wall clock after the levels, the first-column tables, the join, refRight, the traces
  int runF, runR;    // short-read kernel: lane i holds run i of the forward / reverse trace (push order)
};

// The tables live in the wavefront's workspace, which it re-writes for every junction: table reads bypass the vector L1
// (like ld_scratch, split_kernel.hpp), two int16 entries per aligned 32-bit load.
// (the tables live in the wavefront's HBM workspace: the loads are issued as global_load -- a flat_load, which is what a
//  pointer of unknown address space gets, also counts against the LDS counter and makes LDS reads wait for it)
typedef const __attribute__((address_space(1))) uint32_t* sp_gu32;
__device__ __forceinline__ int sp_ld16(const int16_t* p) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(p);
  const uint32_t v = __hip_atomic_load((sp_gu32)(u & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (int)(int16_t)((u & 2) ? (v >> 16) : (v & 0xffffu));
}
__device__ __forceinline__ int sp_ld32(const int32_t* p) {
  return (int)__hip_atomic_load((sp_gu32) reinterpret_cast<uintptr_t>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int sp_ld8(const uint8_t* p) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(p);
  const uint32_t v = __hip_atomic_load((sp_gu32)(u & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (int)((v >> (8 * (u & 3))) & 0xffu);
}

// Where the furthest-reaching tables are read from.  Everything below the level loops -- deep lists, first-column tables,
// refRight, the tracebacks -- goes through get(level, diagonal index q = k + m); "no cell" is any negative value.
//   FrGlobal16: the strip (long-read) kernel's int16 tables in the wavefront's HBM workspace.
//   FrTile8   : the short-read kernel.  Its LDS tile holds the two newest levels (byte rows, row + 1, 0 = none); a level is
//               written to the HBM workspace -- as bytes -- only when the level two above it overwrites it in the tile
//               (sps_level_block), so a junction resolved at S levels has written S - 1 of its S + 1 levels, none at S = 0,
//               and the reads of the newest levels (most of them) cost LDS latency instead of an HBM round trip.
struct FrGlobal16 {
  const int16_t* base;
  int ndp;
  __device__ __forceinline__ int get(int d, int q) const { return sp_ld16(base + (size_t)d * ndp + q); }
};
struct FrTile8 {
  const uint8_t* base;     // [level][ndp] bytes, levels 0 .. S - 2
  int ndp;
  const uint8_t* row0;     // LDS rows of the even / odd newest level, indexed by q (two members, not an array: a
  const uint8_t* row1;     // dynamically indexed array of pointers would live in scratch memory)
  int S;                   // newest level computed
  __device__ __forceinline__ int get(int d, int q) const {
    return (d >= S - 1) ? (int)((d & 1) ? row1 : row0)[q] - 1 : sp_ld8(base + (size_t)d * ndp + q) - 1;
  }
};

__device__ __forceinline__ uint64_t sp_load8(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

__device__ __forceinline__ uint64_t sp_lds8u(const uint8_t* p) {   // (gfx950 reads unaligned LDS quadwords in one instruction)
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// The recurrence of one level (the level loops below carry it out, 64 diagonals per step): diagonal k = q - m at level d
// starts from
//   max( FR[d-1][k],                                  level d - 1 reaches at least as far
//        FR[d-1][k-1]      if that cell's column + 1 <= n,     reference-only move (r, c-1) -> (r, c)
//        FR[d-2][k] + 1    clamped to the diagonal's last row, mismatch (r, c) -> (r+1, c+1)
//        FR[d-2][k+1] + 1  if <= m,                            consensus-only move from diagonal k + 1
//        -k                if k < 0 and 2 |k| <= d,            first column: D[r][0] = 2 r
//        0                 if d == 0 and k >= 0 )              row 0 is free
// and runs down the diagonal while the letters match (8 per compare).  Cells of diagonal k have rows >= -k, so the lower
// column bounds hold for every valid entry; a same-diagonal step past the last row or column can only start from the
// bound itself, which level d - 1 already holds.
// A level of diagonal q only needs the two previous levels of q-1 .. q+1, so a TILE of diagonals can be carried through a
// BLOCK of SP_LB levels entirely in LDS when a halo of SP_LB diagonals on either side is recomputed (the valid region
// shrinks by one diagonal per level and side: 2 x 16 / 960 = 3 % redundant work).  The table rows then cost LDS latency
// instead of an L2 / HBM round trip per level; the global tables are only written (streaming) and read back once per
// block.  The letters a tile can touch -- the whole consensus, window columns [k_lo, k_hi + m] -- are staged in LDS too
// when they fit.
constexpr int SP_LB = 16;                            // levels per block
// TW = valid diagonals per tile, STRCAP = bytes for staged letters (both matrices; 0: the caller's strings are in LDS already),
// ELEM = int16_t (rows as they are; "none" = SP_NEG) or uint8_t (row + 1, 0 = "none": consensus rows <= 254, half the LDS)
template <int TW, int STRCAP, typename ELEM = int16_t, int NROWS = 3>
struct __attribute__((aligned(16))) SpTileT {
  typedef ELEM elem_t;
  static constexpr int tw = TW;
  static constexpr int row_len = TW + 2 * SP_LB + 64;   // halo + slack for the second chunk of an iteration
  static constexpr int str_cap = STRCAP;
  static constexpr bool narrow = sizeof(ELEM) == 1;
  ELEM row[2][NROWS][TW + 2 * SP_LB + 64];              // [matrix][level % NROWS][diagonal - base]
  uint8_t str[STRCAP > 0 ? STRCAP : 8];
};
typedef SpTileT<960, 8192, int16_t, 2> SpTile;       // long reads: 3 % halo overhead, letters staged per tile, two rolling rows

// The level block of the strip (long-read) kernel: tiles of TILE::tw diagonals with a halo of SP_LB, int16 rows, TWO rows
// per matrix (level d overwrites level d - 2 in place: diagonal q reads entries q and q + 1 of that row, chunks ascend,
// every chunk reads before it writes; the shrinking halo only ever reads entries the level two below computed), the
// lean candidate step of sps_level_block.  STAGED: the letters a tile can touch sit in T.str.
template <typename TILE, bool STAGED>
__device__ __forceinline__ void sp_tile_levels(const uint8_t* aF, const uint8_t* bF, const uint8_t* aR, const uint8_t* bR, int c0, int m, int n,
                                               int d0, int nl, int tlo, int thi, int16_t* FRf, int16_t* FRr, int ndp, TILE& T,
                                               int16_t* reachF, int16_t* reachR, int lane) {
  constexpr int OFF = SP_LB + 1;
  const int ND = n + m + 1;
  const int base = tlo - OFF;
  auto ld8 = [&](const uint8_t* p) -> uint64_t { return STAGED ? sp_lds8a(p) : sp_load8(p); };   // (LDS: aligned dwords only, see sp_lds8a)
  for (int j = 0; j < nl; ++j) {
    const int d = d0 + j;
    const int halo = nl - 1 - j;
    const int qa = max(tlo - halo, 0), qb = min(thi + halo, ND);
    int16_t* curF = T.row[0][d & 1] - base;
    int16_t* curR = T.row[1][d & 1] - base;
    const int16_t* p1F = T.row[0][(d + 1) & 1] - base;
    const int16_t* p1R = T.row[1][(d + 1) & 1] - base;
    const int16_t* p2F = curF;
    const int16_t* p2R = curR;
    int16_t* gF = FRf + (size_t)d * ndp;
    int16_t* gR = FRr + (size_t)d * ndp;
    const int seed = (d == 0) ? 0 : -1;
    int rf = -1, rr = -1;
    for (int q0 = qa; q0 < qb; q0 += 2 * WAVE) {
      int bf[2], br[2], kk[2];
      uint64_t zf[2], zr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = min(q0 + u * WAVE + lane, qb - 1);
        const int k = q - m;
        kk[u] = k;
        const int nk = n - k, rmax = min(m, nk);
        const int first = (k < 0 && -2 * k <= d) ? -k : ((k >= 0) ? seed : -1);
        bf[u] = br[u] = first;
        if (d > 0) {
          {
            const int v1 = p1F[q], v1l = p1F[q - 1], v2 = p2F[q], v2r = p2F[q + 1];
            int b = max(v1, first);
            if ((unsigned)v1l <= (unsigned)nk) b = max(b, v1l);          // reference-only move from diagonal k - 1
            if (v2 >= 0) b = max(b, min(v2 + 1, rmax));                   // mismatch on this diagonal
            if (v2r >= 0 && v2r < m) b = max(b, v2r + 1);                 // consensus-only move from diagonal k + 1
            bf[u] = b;
          }
          {
            const int v1 = p1R[q], v1l = p1R[q - 1], v2 = p2R[q], v2r = p2R[q + 1];
            int b = max(v1, first);
            if ((unsigned)v1l <= (unsigned)nk) b = max(b, v1l);
            if (v2 >= 0) b = max(b, min(v2 + 1, rmax));
            if (v2r >= 0 && v2r < m) b = max(b, v2r + 1);
            br[u] = b;
          }
        }
        const int r0 = max(bf[u], 0), r1 = max(br[u], 0);
        zf[u] = ld8(aF + r0) ^ ld8(bF + (max(r0 + k, c0) - c0));
        zr[u] = ld8(aR + r1) ^ ld8(bR + (max(r1 + k, c0) - c0));
      }
      int endF[2], endR[2];
      bool moreF[2], moreR[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = kk[u];
        {
          const int b0 = max(bf[u], 0);
          const int lim = min(m - b0, n - b0 - k);
          const int adv = zf[u] ? (int)(__builtin_ctzll(zf[u]) >> 3) : 8;
          endF[u] = b0 + lim;
          moreF[u] = bf[u] >= 0 && adv >= 8 && lim > 8;
          bf[u] = (bf[u] < 0) ? -1 : b0 + min(adv, lim);
        }
        {
          const int b0 = max(br[u], 0);
          const int lim = min(m - b0, n - b0 - k);
          const int adv = zr[u] ? (int)(__builtin_ctzll(zr[u]) >> 3) : 8;
          endR[u] = b0 + lim;
          moreR[u] = br[u] >= 0 && adv >= 8 && lim > 8;
          br[u] = (br[u] < 0) ? -1 : b0 + min(adv, lim);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u * WAVE + lane, k = kk[u];
        if (__ballot(moreF[u])) {
          if (moreF[u]) {
            int r = bf[u];
            for (;;) {
              const uint64_t z = ld8(aF + r) ^ ld8(bF + (r + k - c0));
              if (z) { r += (int)(__builtin_ctzll(z) >> 3); break; }
              r += 8;
              if (r >= endF[u]) break;
            }
            bf[u] = min(r, endF[u]);
          }
        }
        if (__ballot(moreR[u])) {
          if (moreR[u]) {
            int r = br[u];
            for (;;) {
              const uint64_t z = ld8(aR + r) ^ ld8(bR + (r + k - c0));
              if (z) { r += (int)(__builtin_ctzll(z) >> 3); break; }
              r += 8;
              if (r >= endR[u]) break;
            }
            br[u] = min(r, endR[u]);
          }
        }
        if (q < qb) {
          curF[q] = (int16_t)bf[u];
          curR[q] = (int16_t)br[u];
          if (q >= tlo && q < thi) {
            gF[q] = (int16_t)bf[u];
            gR[q] = (int16_t)br[u];
            rf = max(rf, bf[u]);
            rr = max(rr, br[u]);
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      rf = max(rf, __shfl_xor(rf, o));
      rr = max(rr, __shfl_xor(rr, o));
    }
    if (lane == 0) {
      reachF[d] = (int16_t)max((int)reachF[d], rf < 0 ? SP_NEG : rf);
      reachR[d] = (int16_t)max((int)reachR[d], rr < 0 ? SP_NEG : rr);
    }
    __syncthreads();
  }
}

template <typename TILE>
__device__ DH_SP_FN void sp_level_block2(const uint8_t* consF, const uint8_t* refF, const uint8_t* consR, const uint8_t* refR, int m,
                                             int n, int d0, int d1, int16_t* FRf, int16_t* FRr, int ndp, TILE& T, int16_t* reachF,
                                             int16_t* reachR, int lane) {
  constexpr int SP_TW = TILE::tw;
  constexpr int SP_STR_CAP = TILE::str_cap;
  constexpr int OFF = SP_LB + 1;
  const int ND = n + m + 1;
  const int nl = d1 - d0 + 1;
  for (int d = d0 + lane; d <= d1; d += WAVE) { reachF[d] = (int16_t)SP_NEG; reachR[d] = (int16_t)SP_NEG; }
  __syncthreads();
  const bool stage = 2 * (2 * m + SP_TW + 2 * SP_LB + 64) <= SP_STR_CAP;
  for (int tlo = 0; tlo < ND; tlo += SP_TW) {
    const int thi = min(tlo + SP_TW, ND);
    const int base = tlo - OFF;
    const int len = thi + OFF - base;
    for (int i = lane; i < len; i += WAVE) {     // levels d0 - 1 and d0 - 2 of the tile + halo (level d lives in row d & 1)
      const int q = base + i;
      const bool in = q >= 0 && q < ND;
      T.row[0][(d0 + 1) & 1][i] = (int16_t)((d0 >= 1 && in) ? sp_ld16(FRf + (size_t)(d0 - 1) * ndp + q) : SP_NEG);
      T.row[1][(d0 + 1) & 1][i] = (int16_t)((d0 >= 1 && in) ? sp_ld16(FRr + (size_t)(d0 - 1) * ndp + q) : SP_NEG);
      T.row[0][d0 & 1][i] = (int16_t)((d0 >= 2 && in) ? sp_ld16(FRf + (size_t)(d0 - 2) * ndp + q) : SP_NEG);
      T.row[1][d0 & 1][i] = (int16_t)((d0 >= 2 && in) ? sp_ld16(FRr + (size_t)(d0 - 2) * ndp + q) : SP_NEG);
    }
    if (stage) {
      // letters this tile can touch: rows 0 .. m, columns c = r + k, k in [tlo - SP_LB - m, thi + SP_LB - m)
      const int c0 = max(0, tlo - SP_LB - m) & ~7, c1 = min(n, thi + SP_LB);
      const int wl = max(c1 - c0, 0);
      const int am = (m + 16 + 7) & ~7, bw = (wl + 16 + 7) & ~7;
      uint8_t* la0 = T.str;
      uint8_t* lb0 = la0 + am;
      uint8_t* la1 = lb0 + bw;
      uint8_t* lb1 = la1 + am;
      for (int i = lane; i < am; i += WAVE) { la0[i] = (i < m) ? consF[i] : (uint8_t)1; la1[i] = (i < m) ? consR[i] : (uint8_t)1; }
      for (int i = lane; i < bw; i += WAVE) { lb0[i] = (i < wl) ? refF[c0 + i] : (uint8_t)2; lb1[i] = (i < wl) ? refR[c0 + i] : (uint8_t)2; }
      __syncthreads();
      sp_tile_levels<TILE, true>(la0, lb0, la1, lb1, c0, m, n, d0, nl, tlo, thi, FRf, FRr, ndp, T, reachF, reachR, lane);
    } else {
      __syncthreads();
      sp_tile_levels<TILE, false>(consF, refF, consR, refR, 0, m, n, d0, nl, tlo, thi, FRf, FRr, ndp, T, reachF, reachR, lane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// The level loop again for the short-read kernel: ONE tile that holds every diagonal (no halo to recompute, nothing to
// reload between calls), byte rows (row + 1, 0 = none), the four strings in LDS.  All guards of sp_candidate that a
// "none" entry or a bound already implies are gone (cells of diagonal k have rows >= -k, so the lower column bounds hold
// for every valid entry; a same-diagonal step past the last row or column can only start from the bound itself, which
// level d - 1 already holds): about a third of the instructions of sp_level_block.  TWO rows per matrix: level d is
// written over level d - 2 in place -- diagonal q reads entries q and q + 1 of that row, chunks go in ascending order and
// every chunk reads before it writes, so nothing is overwritten early.  The rows must be zero when d0 == 0 (done here);
// index = diagonal + SP_LB + 1.
#ifdef DH_SPS_DBG
__device__ unsigned long long dh_dbg[16];   // profiling builds: [0] level calls, [1] levels, [2] ticks in the diagonal loop, [3] ticks in the level tail, [4] ticks in the
                                            // block epilogue (store wait), [5] extension calls, [6] ticks in extensions, [7] ticks zeroing the tile
#define DH_DBG_ADD(i, v) do { if (lane == 0) atomicAdd(&dh_dbg[i], (unsigned long long)(v)); } while (0)
#define DH_DBG_T() wall_clock64()
#else
#define DH_DBG_ADD(i, v) do { } while (0)
#define DH_DBG_T() 0ull
#endif
#ifdef DH_SPS_FINE
__shared__ unsigned long long dh_tl[6];   // debug build: wall clock at the start of a level block, after the tile is cleared, after the steps of its last level, at its end
#define DH_TL(i) do { if (lane == 0) dh_tl[i] = wall_clock64(); } while (0)
#else
#define DH_TL(i) do { } while (0)
#endif
constexpr int SPS_OFF = 20;   // index of diagonal 0 in a byte row of the short-read tile (dword aligned; entry -1 is padding)

// ---- the level loop of the short-read kernel, second formulation (round 4) ------------------------------------------------
// Round 3's loop (sps_level_block_v1 below) gave every lane ONE diagonal per step and spent ~110 wave instructions per 64
// table entries, two thirds of them control flow, byte-wide LDS traffic and waits: the loop was issue-bound at four
// wavefronts per SIMD (profiles/r04: 450 instructions per 128 diagonals x 2 matrices, 2.4 cycles each).  Here a lane owns FOUR
// consecutive diagonals: the three tile rows it needs arrive as four dword reads per matrix (levels d - 1 and d - 2 at the
// lane's diagonals, and the same rows shifted by one diagonal -- unaligned dword reads -- for the two neighbour moves), the
// level that leaves the tile goes to the workspace as one coalesced dword store, the new entries are written as one dword.
// The candidate, the first eight-letter compare and the clamp to the diagonal's last row are straight-line code per entry;
// the body is instantiated for the diagonal ranges that need the first-column rule (NEG: k < 0) or the window's right edge
// (EDGE: n - k < m, or diagonals beyond the table) so that the interior pays for neither.  Diagonals whose first compare
// matched all eight letters -- the few an alignment follows -- are finished afterwards with the whole wavefront comparing
// 512 letters at once (one LDS round trip instead of one per eight letters).
// Same recurrence, same table contents as v1 (bit-compared through every sparse parity test; -DDH_SPS_LEVEL_V1 keeps v1).
// four entries of one matrix: diagonals qb .. qb + 3.  e1 / e2: levels d - 1 / d - 2 at those diagonals, e1l: level d - 1 one
// diagonal to the left, e2r: level d - 2 one diagonal to the right (bytes, row + 1, 0 = none).  seed1 = 1 at level 0.
// Returns the four new entries; maxB = running maximum of (row + 1); pend gets bit (pbit + i) for an entry that wants more.
template <bool NEG, bool EDGE>
__device__ __forceinline__ uint32_t sps_entries4(const uint8_t* cons, const uint8_t* ref, uint32_t e1, uint32_t e1l, uint32_t e2, uint32_t e2r,
                                                 int qb, int m, int n, int dhalf, int seed1, int ND, int& maxB, uint32_t& pend, int pbit) {
  // pass 1: the four candidates, then all eight letter loads in flight together; pass 2: the compares
  int Bv[4], r0v[4], rmaxv[4];
  uint64_t zv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = qb + i, k = q - m;
    const int E1 = (int)((e1 >> (8 * i)) & 255u), E1L = (int)((e1l >> (8 * i)) & 255u);
    const int E2 = (int)((e2 >> (8 * i)) & 255u), E2R = (int)((e2r >> (8 * i)) & 255u);
    int rmax = m;
    int B = E1;                                                     // level d - 1 reaches at least as far
    if (EDGE) {
      const int nk = n - k;
      rmax = min(m, nk);
      if ((unsigned)(E1L - 1) <= (unsigned)nk) B = max(B, E1L);     // reference-only move from diagonal k - 1 (its column + 1 <= n)
    } else {
      B = max(B, E1L);
    }
    B = max(B, E2 ? min(E2 + 1, rmax + 1) : 0);                     // mismatch on this diagonal, clamped to its last row
    B = max(B, ((unsigned)(E2R - 1) < (unsigned)m) ? E2R + 1 : 0);  // consensus-only move from diagonal k + 1 (rows 0 .. m - 1)
    if (NEG) {
      const int kk = -k;
      if ((unsigned)(kk - 1) < (unsigned)dhalf) B = max(B, kk + 1); // first column: D[r][0] = 2 r  (k < 0 and 2 |k| <= d)
      B = max(B, (k >= 0) ? seed1 : 0);
    } else {
      B = max(B, seed1);                                            // level 0: row 0 of every diagonal k >= 0
    }
    Bv[i] = B;
    rmaxv[i] = rmax;
    r0v[i] = max(B - 1, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = qb + i - m;
    zv[i] = sp_lds8a(cons + r0v[i]) ^ sp_lds8a(ref + (NEG ? max(r0v[i] + k, 0) : r0v[i] + k));
  }
  uint32_t outw = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = qb + i;
    const int adv = zv[i] ? (int)(__builtin_ctzll(zv[i]) >> 3) : 8;
    int NB = Bv[i] ? min(r0v[i] + adv, rmaxv[i]) + 1 : 0;
    bool more = (Bv[i] != 0) && (adv >= 8) && (rmaxv[i] - r0v[i] > 8);
    if (EDGE) {
      NB = (q < ND) ? NB : 0;
      more = more && (q < ND);
    }
    outw |= (uint32_t)NB << (8 * i);
    maxB = max(maxB, NB);
    pend |= (more ? 1u : 0u) << (pbit + i);
  }
  return outw;
}

// finishes the entries flagged in `pend` (bit i: forward entry i of the lane, bit 4 + i: reverse entry i): for each the whole
// wavefront compares 512 letters from the row the first compare stopped at.  cur rows already hold the provisional entries.
__device__ __forceinline__ void sps_extend_pending(const uint8_t* consF, const uint8_t* refF, const uint8_t* consR, const uint8_t* refR,
                                                   uint8_t* curF, uint8_t* curR, int q0, int m, int n, uint32_t pend, int& maxB_F,
                                                   int& maxB_R, int lane) {
  // Many flagged entries (low-complexity sequence: repeats match eight letters on many diagonals at once): every lane
  // walks its own entries, eight letters per step, all lanes side by side -- the wavefront-wide compare below handles ONE
  // entry per LDS round trip and is the faster way only for the handful an alignment follows.
  int total = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) total += __popcll(__ballot((pend >> e) & 1u));
  if (total > 6) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool mine = (pend >> e) & 1u;
      if (__ballot(mine) == 0ull) continue;
      if (mine) {
        const bool rev = e >= 4;
        const int q = q0 + 4 * lane + (e & 3), k = q - m;
        const uint8_t* cons = rev ? consR : consF;
        const uint8_t* ref = rev ? refR : refF;
        uint8_t* cur = rev ? curR : curF;
        const int end = min(m, n - k);
        int r = (int)cur[q] - 1;
        for (;;) {
          const uint64_t z = sp_lds8a(cons + r) ^ sp_lds8a(ref + r + k);
          if (z) { r += (int)(__builtin_ctzll(z) >> 3); break; }
          r += 8;
          if (r >= end) break;
        }
        r = min(r, end);
        cur[q] = (uint8_t)(r + 1);
        if (rev) maxB_R = max(maxB_R, r + 1); else maxB_F = max(maxB_F, r + 1);
      }
    }
    return;
  }
  unsigned long long lanes = __ballot(pend != 0u);
  while (lanes) {
    const int src = __builtin_ctzll(lanes);
    lanes &= lanes - 1;
    uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)pend, src);
    while (bits) {
      const int e = __builtin_ctz(bits);
      bits &= bits - 1;
      const bool rev = e >= 4;
      const int q = q0 + 4 * src + (e & 3), k = q - m;
      const uint8_t* cons = rev ? consR : consF;
      const uint8_t* ref = rev ? refR : refF;
      uint8_t* cur = rev ? curR : curF;
      const int end = min(m, n - k);
      int r = (int)cur[q] - 1;                          // (uniform address: the provisional entry, row of the first mismatch - none yet)
      for (;;) {
        const int rr = r + 8 * lane;
        uint64_t z = 0;
        if (rr < end) z = sp_lds8a(cons + rr) ^ sp_lds8a(ref + rr + k);
        const unsigned long long stop = __ballot(rr >= end || z != 0ull);
        if (stop == 0ull) { r += 8 * WAVE; continue; }  // (512 matches and still inside: consensus rows <= 254 never get here)
        const int jl = __builtin_ctzll(stop);
        const int rj = r + 8 * jl;
        const uint32_t zlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(z & 0xffffffffull), jl);
        const uint32_t zhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(z >> 32), jl);
        const uint64_t zz = ((uint64_t)zhi << 32) | zlo;
        r = (rj >= end) ? end : min(rj + (zz ? (int)(__builtin_ctzll(zz) >> 3) : 8), end);
        break;
      }
      if (lane == 0) cur[q] = (uint8_t)(r + 1);
      if (rev) maxB_R = max(maxB_R, r + 1); else maxB_F = max(maxB_F, r + 1);
    }
  }
}

template <bool NEG, bool EDGE, bool L0>
__device__ __forceinline__ uint32_t sps_level_step(const uint8_t* consF, const uint8_t* refF, const uint8_t* consR, const uint8_t* refR, int m, int n,
                                               int d, int q0, int ND, uint8_t* curF, uint8_t* curR, const uint8_t* p1F, const uint8_t* p1R,
                                               uint8_t* gF, uint8_t* gR, int& maxB_F, int& maxB_R, int lane) {
  const int qb = q0 + 4 * lane;
  uint32_t e1F = 0, e1lF = 0, e2F = 0, e2rF = 0, e1R = 0, e1lR = 0, e2R = 0, e2rR = 0;
  if (!L0) {
    // aligned dwords only: the rows shifted by one diagonal are composed from the neighbouring dword (v_alignbyte_b32)
    const uint32_t* a1F = reinterpret_cast<const uint32_t*>(p1F + qb);
    const uint32_t* a2F = reinterpret_cast<const uint32_t*>(curF + qb);
    const uint32_t* a1R = reinterpret_cast<const uint32_t*>(p1R + qb);
    const uint32_t* a2R = reinterpret_cast<const uint32_t*>(curR + qb);
    e1F = a1F[0]; e1lF = __builtin_amdgcn_alignbyte(e1F, a1F[-1], 3u); e2F = a2F[0]; e2rF = __builtin_amdgcn_alignbyte(a2F[1], e2F, 1u);
    e1R = a1R[0]; e1lR = __builtin_amdgcn_alignbyte(e1R, a1R[-1], 3u); e2R = a2R[0]; e2rR = __builtin_amdgcn_alignbyte(a2R[1], e2R, 1u);
    if (d >= 2 && qb < ND) {   // level d - 2 leaves the tile now (level d is written over it): to the workspace as bytes
      *reinterpret_cast<uint32_t*>(gF + qb) = e2F;
      *reinterpret_cast<uint32_t*>(gR + qb) = e2R;
    }
  }
  uint32_t pend = 0;
  const int seed1 = L0 ? 1 : 0, dhalf = d >> 1;
  const uint32_t oF = sps_entries4<NEG, EDGE>(consF, refF, e1F, e1lF, e2F, e2rF, qb, m, n, dhalf, seed1, ND, maxB_F, pend, 0);
  __builtin_amdgcn_sched_barrier(0);   // (the two matrices one after the other: eight 64-bit letter words in flight, not sixteen -- no spills at 128 registers)
  const uint32_t oR = sps_entries4<NEG, EDGE>(consR, refR, e1R, e1lR, e2R, e2rR, qb, m, n, dhalf, seed1, ND, maxB_R, pend, 4);
  if (qb < ND) {
    *reinterpret_cast<uint32_t*>(curF + qb) = oF;
    *reinterpret_cast<uint32_t*>(curR + qb) = oR;
  }
  return pend;   // entries whose first compare matched all eight letters: finished by the caller (ONE copy of that code)
}

template <typename TILE>
__device__ DH_SP_FN void sps_level_block(const uint8_t* consF, const uint8_t* refF, const uint8_t* consR, const uint8_t* refR, int m,
                                         int n, int d0, int d1, int16_t* FRf, int16_t* FRr, int ndp, TILE& T, int16_t* reachF,
                                         int16_t* reachR, int lane) {
  static_assert(TILE::narrow, "byte rows");
  static_assert(sizeof(T.row[0][0]) % 4 == 0 && SPS_OFF % 4 == 0, "dword access to the tile rows");
  const int ND = n + m + 1;
  const unsigned long long tz0 = DH_DBG_T();
  DH_TL(0);
  if (d0 == 0) {
    static_assert(sizeof(T.row) % 16 == 0 && alignof(TILE) >= 16, "the tile is cleared sixteen bytes per lane and store");
    uint4* z = reinterpret_cast<uint4*>(&T.row[0][0][0]);
    for (int i = lane; i < (int)(sizeof(T.row) / 16); i += WAVE) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  DH_TL(1);
  DH_DBG_ADD(0, 1);
  DH_DBG_ADD(7, DH_DBG_T() - tz0);
  for (int d = d0; d <= d1; ++d) {
    const unsigned long long tl0 = DH_DBG_T();
    uint8_t* curF = T.row[0][d & 1] + SPS_OFF;
    uint8_t* curR = T.row[1][d & 1] + SPS_OFF;
    const uint8_t* p1F = T.row[0][(d + 1) & 1] + SPS_OFF;
    const uint8_t* p1R = T.row[1][(d + 1) & 1] + SPS_OFF;
    uint8_t* gF = reinterpret_cast<uint8_t*>(FRf) + (size_t)max(d - 2, 0) * ndp;
    uint8_t* gR = reinterpret_cast<uint8_t*>(FRr) + (size_t)max(d - 2, 0) * ndp;
    int maxB_F = 0, maxB_R = 0;
    for (int q0 = 0; q0 < ND; q0 += 4 * WAVE) {
      const bool neg = q0 < m;                          // some diagonal of the step lies below the main one
      const bool edge = q0 + 4 * WAVE - 1 > n;          // some diagonal has fewer than m rows inside the window, or lies beyond the table
      uint32_t pend;
      if (d == 0) {
        if (neg || edge) pend = sps_level_step<true, true, true>(consF, refF, consR, refR, m, n, d, q0, ND, curF, curR, p1F, p1R, gF, gR, maxB_F, maxB_R, lane);
        else pend = sps_level_step<false, false, true>(consF, refF, consR, refR, m, n, d, q0, ND, curF, curR, p1F, p1R, gF, gR, maxB_F, maxB_R, lane);
      } else if (neg || edge) {
        pend = sps_level_step<true, true, false>(consF, refF, consR, refR, m, n, d, q0, ND, curF, curR, p1F, p1R, gF, gR, maxB_F, maxB_R, lane);
      } else {
        pend = sps_level_step<false, false, false>(consF, refF, consR, refR, m, n, d, q0, ND, curF, curR, p1F, p1R, gF, gR, maxB_F, maxB_R, lane);
      }
      if (__ballot(pend != 0u)) {
        const unsigned long long te = DH_DBG_T();
        sps_extend_pending(consF, refF, consR, refR, curF, curR, q0, m, n, pend, maxB_F, maxB_R, lane);
        DH_DBG_ADD(5, 1);
        DH_DBG_ADD(6, DH_DBG_T() - te);
      }
    }
    const unsigned long long tl1 = DH_DBG_T();
    DH_TL(2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      maxB_F = max(maxB_F, __shfl_xor(maxB_F, o));
      maxB_R = max(maxB_R, __shfl_xor(maxB_R, o));
    }
    if (lane == 0) {
      reachF[d] = (int16_t)(maxB_F <= 0 ? SP_NEG : maxB_F - 1);
      reachR[d] = (int16_t)(maxB_R <= 0 ? SP_NEG : maxB_R - 1);
    }
    __syncthreads();
    DH_DBG_ADD(1, 1);
    DH_DBG_ADD(2, tl1 - tl0);
    DH_DBG_ADD(3, DH_DBG_T() - tl1);
  }
  const unsigned long long tw0 = DH_DBG_T();
  // (levels that left the tile are read back by the traces: their stores first.  Without such a level -- every junction resolved
  //  at S = 0, half of BASELINE's -- this would only wait for the acknowledgements of the set-up stage's stores, the record
  //  defaults and the consensus copy: microseconds under load)
  if (d1 >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DH_TL(3);
  DH_DBG_ADD(4, DH_DBG_T() - tw0);
}

template <typename TILE>
__device__ DH_SP_FN void sps_level_block_v1(const uint8_t* consF, const uint8_t* refF, const uint8_t* consR, const uint8_t* refR, int m,
                                             int n, int d0, int d1, int16_t* FRf, int16_t* FRr, int ndp, TILE& T, int16_t* reachF,
                                             int16_t* reachR, int lane) {
  static_assert(TILE::narrow, "byte rows");
  constexpr int OFF = SPS_OFF;
  const int ND = n + m + 1;
  if (d0 == 0) {
    uint32_t* z = reinterpret_cast<uint32_t*>(&T.row[0][0][0]);
    for (int i = lane; i < (int)(sizeof(T.row) / 4); i += WAVE) z[i] = 0u;
  }
  __syncthreads();
  for (int d = d0; d <= d1; ++d) {
    uint8_t* curF = T.row[0][d & 1] + OFF;
    uint8_t* curR = T.row[1][d & 1] + OFF;
    const uint8_t* p1F = T.row[0][(d + 1) & 1] + OFF;
    const uint8_t* p2F = curF;
    const uint8_t* p1R = T.row[1][(d + 1) & 1] + OFF;
    const uint8_t* p2R = curR;
    // level d - 2 leaves the tile now (level d is written over it): its entries go to the workspace as bytes
    uint8_t* gF = reinterpret_cast<uint8_t*>(FRf) + (size_t)max(d - 2, 0) * ndp;
    uint8_t* gR = reinterpret_cast<uint8_t*>(FRr) + (size_t)max(d - 2, 0) * ndp;
    const int seed = (d == 0) ? 0 : -1;        // level 0: row 0 of every diagonal k >= 0
    int rf = -1, rr = -1;
    for (int q0 = 0; q0 < ND; q0 += 2 * WAVE) {
      int bf[2], br[2], kk[2];
      uint64_t zf[2], zr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = min(q0 + u * WAVE + lane, ND - 1);
        const int k = q - m;
        kk[u] = k;
        const int nk = n - k, rmax = min(m, nk);
        const int first = (k < 0 && -2 * k <= d) ? -k : ((k >= 0) ? seed : -1);
        bf[u] = br[u] = first;
        if (d > 0) {     // (level 0 has no predecessors: row 0 of the diagonals k >= 0)
          if (d >= 2 && q0 + u * WAVE + lane < ND) {
            gF[q] = p2F[q];
            gR[q] = p2R[q];
          }
          {
            const int e1 = p1F[q], e1l = p1F[q - 1], e2 = p2F[q], e2r = p2F[q + 1];
            int b = max(e1 - 1, first);
            if ((unsigned)(e1l - 1) <= (unsigned)nk) b = max(b, e1l - 1);      // reference-only move from diagonal k - 1
            if (e2 >= 1) b = max(b, min(e2, rmax));                            // mismatch on this diagonal
            if (e2r >= 1 && e2r <= m) b = max(b, e2r);                         // consensus-only move from diagonal k + 1
            bf[u] = b;
          }
          {
            const int e1 = p1R[q], e1l = p1R[q - 1], e2 = p2R[q], e2r = p2R[q + 1];
            int b = max(e1 - 1, first);
            if ((unsigned)(e1l - 1) <= (unsigned)nk) b = max(b, e1l - 1);
            if (e2 >= 1) b = max(b, min(e2, rmax));
            if (e2r >= 1 && e2r <= m) b = max(b, e2r);
            br[u] = b;
          }
        }
        const int r0 = max(bf[u], 0), r1 = max(br[u], 0);
        zf[u] = sp_lds8u(consF + r0) ^ sp_lds8u(refF + max(r0 + k, 0));
        zr[u] = sp_lds8u(consR + r1) ^ sp_lds8u(refR + max(r1 + k, 0));
      }
      // straight-line first compare of the four streams (a "none" entry compares row 0 and is discarded); only a run of
      // 8 or more matches -- the few diagonals an alignment follows -- enters the loop below
      int endF[2], endR[2];
      bool moreF[2], moreR[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = kk[u];
        {
          const int b0 = max(bf[u], 0);
          const int lim = min(m - b0, n - b0 - k);
          const int adv = zf[u] ? (int)(__builtin_ctzll(zf[u]) >> 3) : 8;
          endF[u] = b0 + lim;
          moreF[u] = bf[u] >= 0 && adv >= 8 && lim > 8;
          bf[u] = (bf[u] < 0) ? -1 : b0 + min(adv, lim);
        }
        {
          const int b0 = max(br[u], 0);
          const int lim = min(m - b0, n - b0 - k);
          const int adv = zr[u] ? (int)(__builtin_ctzll(zr[u]) >> 3) : 8;
          endR[u] = b0 + lim;
          moreR[u] = br[u] >= 0 && adv >= 8 && lim > 8;
          br[u] = (br[u] < 0) ? -1 : b0 + min(adv, lim);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u * WAVE + lane, k = kk[u];
        if (__ballot(moreF[u])) {
          if (moreF[u]) {
            int r = bf[u];
            for (;;) {
              const uint64_t z = sp_lds8u(consF + r) ^ sp_lds8u(refF + r + k);
              if (z) { r += (int)(__builtin_ctzll(z) >> 3); break; }
              r += 8;
              if (r >= endF[u]) break;
            }
            bf[u] = min(r, endF[u]);
          }
        }
        if (__ballot(moreR[u])) {
          if (moreR[u]) {
            int r = br[u];
            for (;;) {
              const uint64_t z = sp_lds8u(consR + r) ^ sp_lds8u(refR + r + k);
              if (z) { r += (int)(__builtin_ctzll(z) >> 3); break; }
              r += 8;
              if (r >= endR[u]) break;
            }
            br[u] = min(r, endR[u]);
          }
        }
        if (q < ND) {
          curF[q] = (uint8_t)(bf[u] + 1);
          curR[q] = (uint8_t)(br[u] + 1);
          rf = max(rf, bf[u]);
          rr = max(rr, br[u]);
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      rf = max(rf, __shfl_xor(rf, o));
      rr = max(rr, __shfl_xor(rr, o));
    }
    if (lane == 0) {
      reachF[d] = (int16_t)(rf < 0 ? SP_NEG : rf);
      reachR[d] = (int16_t)(rr < 0 ? SP_NEG : rr);
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// First columns of one level without same-address atomics.  Diagonals k >= 0 come in ascending order and all start at
// row 0, so diagonal k answers exactly the rows above everything the earlier ones reached: a running maximum (pm, uniform)
// plus a prefix maximum over the 64 lanes of a chunk gives every lane its own, disjoint row range.
__device__ __forceinline__ void sp_offer_chunk(int32_t* row, int rlo, int rhi, int k, int v, int& pm, int lane) {
  if (__ballot(v > pm && v >= rlo) == 0ull) return;   // (no diagonal of the chunk reaches beyond the rows already answered: the usual case)
  int incl = v;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl = max(incl, t);
  }
  int excl = __shfl_up(incl, 1);
  if (lane == 0) excl = SP_NEG;
  const int from = max(max(excl, pm) + 1, rlo), to = min(v, rhi);
  const int hs = min(to, from + 7);
  for (int r = from; r <= hs; ++r) atomicMin(&row[r], r + k);
  unsigned long long todo = __ballot(to > from + 7);     // (the few diagonals the alignment runs along: whole wavefront)
  while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int f2 = __shfl(from, src) + 8, t2 = __shfl(to, src), k2 = __shfl(k, src);
    for (int r = f2 + lane; r <= t2; r += WAVE) atomicMin(&row[r], r + k2);
  }
  pm = max(pm, __shfl(incl, WAVE - 1));
}
// Diagonals below the main one start at (row -k, column 0), a cell of deficit 2 |k|: at level d only the d / 2 diagonals
// next to the main one hold cells.  Each offers its rows with the whole wavefront.
template <typename FRV>
__device__ __forceinline__ void sp_offer_negative(const FRV& FR, int32_t* row, int m, int d, int rlo, int rhi, int lane) {
  const int nneg = min(d / 2, m);
  for (int i0 = 0; i0 < nneg; i0 += WAVE) {
    const int i = i0 + lane;
    const int v = (i < nneg) ? FR.get(d, m - 1 - i) : SP_NEG;
    const int cnt = min(WAVE, nneg - i0);
    for (int t = 0; t < cnt; ++t) {
      const int vt = __shfl(v, t), kt = -(i0 + t + 1);
      const int lo = max(rlo, -kt), hi = min(vt, rhi);
      for (int r = lo + lane; r <= hi; r += WAVE) atomicMin(&row[r], r + kt);
    }
  }
}

// cT[d][r] = min over diagonals k with FR[d][k] >= r (and r on the diagonal) of r + k, for rows rlo .. rhi, levels 0 .. S
template <typename FRV>
__device__ DH_SP_FN void sp_first_columns(const FRV& FR, int m, int n, int S, int rlo, int rhi, int32_t* cT, int lane) {
  const int ND = n + m + 1;
  const int rows = rhi - rlo + 1;
  for (int d = 0; d <= S; ++d)
    for (int i = lane; i < rows; i += WAVE) cT[(size_t)d * (m + 1) + rlo + i] = SP_INF;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // four levels at a time: their table loads are in flight together (a wide row range -- a consensus that aligns without a
  // split -- makes this the longest phase of a junction, and one L2 round trip per chunk and level was most of it)
  for (int d0 = 0; d0 <= S; d0 += 4) {
    int pm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pm[u] = (d0 + u <= S) ? rlo - 1 : rhi;
      if (d0 + u <= S) sp_offer_negative(FR, cT + (size_t)(d0 + u) * (m + 1), m, d0 + u, rlo, rhi, lane);
    }
    for (int q0 = m; q0 < ND && (pm[0] < rhi || pm[1] < rhi || pm[2] < rhi || pm[3] < rhi); q0 += WAVE) {
      const int q = q0 + lane;
      int v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (q < ND && pm[u] < rhi) ? FR.get(min(d0 + u, S), q) : SP_NEG;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (pm[u] < rhi) sp_offer_chunk(cT + (size_t)(d0 + u) * (m + 1), rlo, rhi, q - m, v[u], pm[u], lane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// the diagonals that reach row rlo at level S (FR is non-decreasing in the level: no other diagonal can reach it at a lower
// one), compacted into list[]; returns their number or -1 when the list overflows
template <typename FRV>
__device__ DH_SP_FN int sp_deep_list(const FRV& FR, int m, int n, int S, int rlo, int32_t* list, int cap, int lane) {
  const int ND = n + m + 1;
  int cnt = 0;
  for (int q0 = 0; q0 < ND; q0 += WAVE) {
    const int q = q0 + lane;
    const bool hit = (q < ND) && (FR.get(S, min(q, ND - 1)) >= rlo);
    const unsigned long long bm = __ballot(hit);
    if (bm) {
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pos = cnt + __popcll(bm & below);
      if (hit && pos < cap) list[pos] = q;
      cnt += __popcll(bm);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return (cnt <= cap) ? cnt : -1;
}

// sp_deep_list on the byte row of level S the short-read tile still holds (row + 1, 0 = none; lv = row + SPS_OFF)
template <bool WAIT = true>
__device__ __forceinline__ int sps_deep_list(const uint8_t* lv, int ND, int rlo, int32_t* list, int cap, int lane) {
  int cnt = 0;
  for (int q0 = 0; q0 < ND; q0 += WAVE) {
    const int q = q0 + lane;
    const bool hit = (q < ND) && ((int)lv[min(q, ND - 1)] - 1 >= rlo);
    const unsigned long long bm = __ballot(hit);
    if (bm) {
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pos = cnt + __popcll(bm & below);
      if (hit && pos < cap) list[pos] = q;
      cnt += __popcll(bm);
    }
  }
  if (WAIT) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  return (cnt <= cap) ? cnt : -1;
}

// sp_first_columns restricted to the listed diagonals (ascending)
template <typename FRV>
__device__ DH_SP_FN void sp_first_columns_list(const FRV& FR, int m, int S, int rlo, int rhi, const int32_t* list,
                                                   int cnt, int32_t* cT, int lane) {
  const int rows = rhi - rlo + 1;
  for (int d = 0; d <= S; ++d)
    for (int i = lane; i < rows; i += WAVE) cT[(size_t)d * (m + 1) + rlo + i] = SP_INF;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (cnt <= WAVE) {       // the usual case: one chunk, its entries stay in registers, four levels' loads in flight
    const int q = (lane < cnt) ? sp_ld32(list + lane) : -1;
    const bool use = q >= m;                     // (diagonals below the main one: sp_offer_negative)
    for (int d0 = 0; d0 <= S; d0 += 4) {
      int v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (use && d0 + u <= S) ? FR.get(d0 + u, q) : SP_NEG;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (d0 + u <= S) {
          int pm = rlo - 1;
          sp_offer_chunk(cT + (size_t)(d0 + u) * (m + 1), rlo, rhi, q - m, v[u], pm, lane);
        }
      }
    }
  } else {
    for (int d = 0; d <= S; ++d) {
      int pm = rlo - 1;
      for (int i0 = 0; i0 < cnt && pm < rhi; i0 += WAVE) {
        const int i = i0 + lane;
        const int q = (i < cnt) ? sp_ld32(list + i) : -1;
        const int v = (q >= m) ? FR.get(d, q) : SP_NEG;
        sp_offer_chunk(cT + (size_t)d * (m + 1), rlo, rhi, q - m, v, pm, lane);
      }
    }
  }
  for (int d = 2; d <= S; ++d) sp_offer_negative(FR, cT + (size_t)d * (m + 1), m, d, rlo, rhi, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Both first-column tables of the short-read kernel in one go when each deep list fits one chunk (the usual case): the
// stores that clear the tables and write the lists are waited for ONCE, the list entries and the table values of four
// levels of BOTH matrices are in flight together -- two memory round trips instead of six.
template <typename FRV>
__device__ DH_SP_FN void sp_first_columns_both(const FRV& FRf, const FRV& FRr, int m, int S, int rloF, int rhiF,
                                                   const int32_t* listF, int cntF, int32_t* cF, int rloR, int rhiR,
                                                   const int32_t* listR, int cntR, int32_t* cR, int lane) {
  const int rows = rhiF - rloF + 1;   // (= rhiR - rloR + 1)
  for (int d = 0; d <= S; ++d)
    for (int i = lane; i < rows; i += WAVE) {
      cF[(size_t)d * (m + 1) + rloF + i] = SP_INF;
      cR[(size_t)d * (m + 1) + rloR + i] = SP_INF;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int qF = (lane < cntF) ? sp_ld32(listF + lane) : -1;
  const int qR = (lane < cntR) ? sp_ld32(listR + lane) : -1;
  const bool useF = qF >= m, useR = qR >= m;   // (diagonals below the main one: sp_offer_negative)
  for (int d0 = 0; d0 <= S; d0 += 4) {
    int vF[4], vR[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vF[u] = (useF && d0 + u <= S) ? FRf.get(d0 + u, qF) : SP_NEG;
      vR[u] = (useR && d0 + u <= S) ? FRr.get(d0 + u, qR) : SP_NEG;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (d0 + u <= S) {
        int pmF = rloF - 1, pmR = rloR - 1;
        sp_offer_chunk(cF + (size_t)(d0 + u) * (m + 1), rloF, rhiF, qF - m, vF[u], pmF, lane);
        sp_offer_chunk(cR + (size_t)(d0 + u) * (m + 1), rloR, rhiR, qR - m, vR[u], pmR, lane);
      }
    }
  }
  for (int d = 2; d <= S; ++d) {
    sp_offer_negative(FRf, cF + (size_t)d * (m + 1), m, d, rloF, rhiF, lane);
    sp_offer_negative(FRr, cR + (size_t)d * (m + 1), m, d, rloR, rhiR, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// traceback from (r, c) with deficit D: the reference's rule (vertical, then horizontal, then diagonal; src/needle.h:154-192)
// decided on the tables; runs in push order.  Returns the number of runs or -1 on overflow.  Wave-uniform.
// REGRUNS: the runs stay in a register -- lane i holds run i (at most WAVE runs; more: overflow) -- instead of going through
// the wavefront's HBM workspace (the short-read kernel: nothing of a junction resolved at <= 1 level touches the workspace)
template <typename FRV, bool REGRUNS = false>
__device__ DH_SP_FN int sp_trace(const FRV& FR, int m, int n, int r, int c, int D, int32_t* runs, int cap, int lane,
                                     int* mismatches = nullptr, int* runreg = nullptr) {
  const int ND = n + m + 1;
  const int D0 = rfl(D);
  int nv = 0, nh = 0;
  auto fr = [&](int d, int q) -> int { return (d < 0 || q < 0 || q >= ND) ? SP_NEG : FR.get(d, q); };
  int nruns = 0, last_op = -1, last_len = 0;
  auto emit = [&](int op, int len) {
    if (len <= 0) return;
    if (op == last_op) { last_len += len; return; }
    if (last_op >= 0) {
      if (REGRUNS) { if (lane == nruns) *runreg = (last_op << 24) | last_len; }
      else if (nruns < cap && lane == 0) runs[nruns] = (last_op << 24) | last_len;
      ++nruns;
    }
    last_op = op;
    last_len = len;
  };
  r = rfl(r); c = rfl(c); D = rfl(D);
  while (r > 0 || c > 0) {
    if (r == 0) { emit(2, c); c = 0; break; }            // (leading reference letters: free)
    if (c == 0) { emit(1, r); nv += r; r = 0; break; }
    const int k = c - r, q = k + m;
    if (D >= 2 && rfl(fr(D - 2, q + 1)) >= r - 1) { emit(1, 1); ++nv; --r; D -= 2; continue; }
    if (D >= 1 && rfl(fr(D - 1, q - 1)) >= r) { emit(2, 1); ++nh; --c; D -= 1; continue; }
    // diagonal run at cost D: rows >= lowD of this diagonal cost D; a test passes at rows <= T
    const int lowD = (D >= 1) ? max(rfl(fr(D - 1, q)) + 1, 0) : 0;
    int T = SP_NEG;
    if (D >= 2) T = max(T, rfl(fr(D - 2, q + 1)) + 1);
    if (D >= 1) T = max(T, rfl(fr(D - 1, q - 1)));
    const int start = max(0, -k);                       // first row of the diagonal (c = 0 or r = 0 there)
    const int y = max(max(lowD, T + 1), start + 1);     // the lowest row the run still leaves diagonally
    const int cnt = r - y + 1;                          // >= 1: the tests failed at row r
    emit(0, cnt);
    r -= cnt;
    c -= cnt;
    if (r < lowD) {                                     // crossed a mismatch: the cost of the cell we stand on now
      int e = D - 1;
      while (e >= 1 && rfl(fr(e - 1, q)) >= r) --e;
      D = (rfl(fr(e, q)) >= r) ? e : D;                 // (e >= 0 always holds a cell here; defensive)
    }
  }
  if (last_op >= 0) {
    if (REGRUNS) { if (lane == nruns) *runreg = (last_op << 24) | last_len; }
    else if (nruns < cap && lane == 0) runs[nruns] = (last_op << 24) | last_len;
    ++nruns;
  }
  if (mismatches) *mismatches = (D0 - 2 * nv - nh) / 2;
  if (!REGRUNS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  return (nruns <= (REGRUNS ? WAVE : cap)) ? nruns : -1;
}

// ---- evaluation of the short-read kernel without the workspace (round 4) ---------------------------------------------------
// Round 3 built the deep lists, the first-column tables and the run lists in the wavefront's HBM workspace: ~30 dependent
// global loads / atomics per junction, each a round trip of 1 - 2 us under load, which is what the phases after the level
// loop were made of (profiles/r04).  When both deep lists fit one lane each (<= 64 diagonals, the usual case: 1 - 20) and the
// levels in question are <= 8, nothing needs memory: lane t keeps its diagonal and that diagonal's furthest rows at levels
// 0 .. SE as packed bytes; the first column of row r at level d,
//     cF[d][r] = r + min { k_t : FR[d][k_t] >= r, r >= -k_t },
// is a loop over the listed diagonals with v_readlane (rows live one per lane), and the join proceeds as before.
// State left for refRight: the reverse list (qR, packed rows) stays in registers.
struct SpsSmall {
  int qF, qR, nF, nR;         // lane t: its listed diagonal index (q = k + m), -1 beyond the list; list lengths
  uint32_t pF[3], pR[3];      // rows + 1 of that diagonal at levels 0 .. 8, one byte each (0 = none)
};

template <typename FRV>
__device__ __forceinline__ bool sps_lists_small(const FRV& frF, const FRV& frR, const uint8_t* lvF, const uint8_t* lvR, int ND, int m, int SE,
                                                int rlo, int rloR, SpsSmall& Q, int lane) {
  Q.qF = Q.qR = -1;
  int nF = 0, nR = 0;
  // four table entries per lane and LDS round trip (the rows are dword aligned, entries at and beyond ND are 0 = "none"); a step
  // without a deep diagonal -- most of them: an alignment follows 1 - 20 diagonals -- costs one ballot.  The order of the list
  // does not matter (the join takes minima over it, refRight a maximum).
  for (int q0 = 0; q0 < ND; q0 += 4 * WAVE) {
    const int qb = q0 + 4 * lane;
    const uint32_t wf = *reinterpret_cast<const uint32_t*>(lvF + qb), wr = *reinterpret_cast<const uint32_t*>(lvR + qb);
    uint32_t hf = 0u, hr = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = qb + i < ND;
      hf |= (uint32_t)(in && (int)((wf >> (8 * i)) & 255u) - 1 >= rlo) << i;
      hr |= (uint32_t)(in && (int)((wr >> (8 * i)) & 255u) - 1 >= rloR) << i;
    }
    if (__ballot((hf | hr) != 0u) == 0ull) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned long long bf = __ballot((hf >> i) & 1u), br = __ballot((hr >> i) & 1u);
      if (nF + __popcll(bf) > WAVE || nR + __popcll(br) > WAVE) return false;
      while (bf) {
        const int b = __builtin_ctzll(bf);
        bf &= bf - 1;
        Q.qF = (lane == nF) ? q0 + 4 * b + i : Q.qF;
        ++nF;
      }
      while (br) {
        const int b = __builtin_ctzll(br);
        br &= br - 1;
        Q.qR = (lane == nR) ? q0 + 4 * b + i : Q.qR;
        ++nR;
      }
    }
  }
  Q.nF = nF;
  Q.nR = nR;
#pragma unroll
  for (int w = 0; w < 3; ++w) { Q.pF[w] = 0u; Q.pR[w] = 0u; }
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    if (d <= SE) {
      const int vf = (Q.qF >= 0) ? frF.get(d, Q.qF) + 1 : 0;
      const int vr = (Q.qR >= 0) ? frR.get(d, Q.qR) + 1 : 0;
      Q.pF[d >> 2] |= (uint32_t)(vf & 255) << (8 * (d & 3));
      Q.pR[d >> 2] |= (uint32_t)(vr & 255) << (8 * (d & 3));
    }
  }
  return true;
}

// One row of the join with at most nine levels in registers (needle.h:96-115): lo_[d] = first column the forward side reaches
// with d deficits, c2_[q] = first column of the reverse side with q.  The minimal e for a d is the number of levels whose reverse
// column is still too far right.  Written without branches: as nested ifs over per-lane data this compiled to exec-mask trees.
// NL: the levels 0 .. SE fit NL slots (the loops are unrolled over NL, not over nine: at SE = 0 -- half of BASELINE's junctions --
// the 81 comparisons of the general form were 80 too many)
template <int NL>
__device__ __forceinline__ void sps_row_best(const int (&lo_)[9], const int (&c2_)[9], int SE, int n, int r, long long& kb, int& db) {
  int thr[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) thr[q] = (q <= SE) ? ((c2_[q] > n) ? -1 : n - c2_[q]) : SP_INF;   // level q is "too far" iff lo > thr[q]
#pragma unroll
  for (int d = 0; d < NL; ++d) {
    const int lo = lo_[d];
    int e = 0;
#pragma unroll
    for (int q = 0; q < NL; ++q) e += (int)(lo > thr[q]);
    const long long kk = ((long long)(d + e) << 40) | ((long long)r << 20) | (long long)lo;
    const bool take = (bool)((int)(d <= SE) & (int)(lo <= n) & (int)(e <= SE) & (int)(d + e <= SE) & (int)(kk <= kb));
    kb = take ? kk : kb;
    db = take ? d : db;
  }
}

// join over the rows rlo .. rhi (needle.h:96-115) from the registers of sps_lists_small: key = (total deficit << 40) | (row << 20)
// | column of the winner (0x7fff... : none), dsel = its forward deficit.  Same arithmetic as the table version below.
template <int NL>
__device__ __forceinline__ void sps_join_small_t(const SpsSmall& Q, int m, int n, int SE, int rlo, int rhi, long long& key, int& dsel, int lane) {
  key = 0x7fffffffffffffffll;
  int dbest = 0;
  for (int r0 = rlo; r0 <= rhi; r0 += WAVE) {
    const int r = r0 + lane, rr = m - r;       // forward row, reverse row
    int lo_[9], c2_[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) { lo_[d] = SP_INF; c2_[d] = SP_INF; }
    for (int t = 0; t < Q.nF; ++t) {
      const int k = __builtin_amdgcn_readlane(Q.qF, t) - m;
      const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pF[0], t), p1 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pF[1], t),
                     p2 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pF[2], t);
      const bool on = r + k >= 0;              // the row exists on this diagonal (k < 0: rows >= -k)
#pragma unroll
      for (int d = 0; d < NL; ++d) {
        if (d <= SE) {
          const int v = (int)((((d < 4) ? p0 : (d < 8) ? p1 : p2) >> (8 * (d & 3))) & 255u) - 1;
          lo_[d] = min(lo_[d], ((int)on & (int)(v >= r)) ? r + k : SP_INF);   // (`on && ..` came out as divergent branches)
        }
      }
    }
    for (int t = 0; t < Q.nR; ++t) {
      const int k = __builtin_amdgcn_readlane(Q.qR, t) - m;
      const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pR[0], t), p1 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pR[1], t),
                     p2 = (uint32_t)__builtin_amdgcn_readlane((int)Q.pR[2], t);
      const bool on = rr + k >= 0;
#pragma unroll
      for (int d = 0; d < NL; ++d) {
        if (d <= SE) {
          const int v = (int)((((d < 4) ? p0 : (d < 8) ? p1 : p2) >> (8 * (d & 3))) & 255u) - 1;
          c2_[d] = min(c2_[d], ((int)on & (int)(v >= rr)) ? rr + k : SP_INF);
        }
      }
    }
    if (r <= rhi) {
      long long kb = 0x7fffffffffffffffll;
      int db = 0;
      sps_row_best<NL>(lo_, c2_, SE, n, r, kb, db);
      if (kb < key) { key = kb; dbest = db; }
    }
  }
  long long kmin = key;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int lo = __shfl_xor((int)(kmin & 0xffffffffll), o), hi = __shfl_xor((int)(kmin >> 32), o);
    const long long w = ((long long)hi << 32) | (unsigned int)lo;
    kmin = w < kmin ? w : kmin;
  }
  const unsigned long long who = __ballot(key == kmin && key != 0x7fffffffffffffffll);
  dsel = who ? __shfl(dbest, __builtin_ctzll(who)) : 0;
  key = kmin;
}
__device__ __forceinline__ void sps_join_small(const SpsSmall& Q, int m, int n, int SE, int rlo, int rhi, long long& key, int& dsel, int lane) {
  if (SE <= 0) sps_join_small_t<1>(Q, m, n, SE, rlo, rhi, key, dsel, lane);
  else if (SE <= 2) sps_join_small_t<3>(Q, m, n, SE, rlo, rhi, key, dsel, lane);
  else sps_join_small_t<9>(Q, m, n, SE, rlo, rhi, key, dsel, lane);
}

// The whole procedure for one junction (one wavefront).  cons / ref: clean letters; rcons / rref: their reverse
// complements.  reach arrays in LDS (SP_LEVELS_MAX int16 each).
template <typename TILE, bool LDSSTR>
__device__ DH_SP_FN SparseRes sparse_long_needle(const uint8_t* cons, const uint8_t* rcons, const uint8_t* ref, const uint8_t* rref,
                                                    int m, int n, const SparseWs& W, TILE& T, int16_t* reachF, int16_t* reachR,
                                                    int s_first, int lane) {
  SparseRes O;
  O.resolved = 0; O.found = 0; O.unsplit = SP_UNKNOWN; O.best = 0; O.consLeft = O.refLeft = O.refRight = 0; O.nrunsF = O.nrunsR = 0; O.levels = 0; O.mmF = O.mmR = 0;
  const int ND = n + m + 1;
  if (ND + 1 > W.ndp || m < 1 || n < 1 || W.smax < 4) return O;
  int done = -1;           // levels 0 .. done are computed
  int S = min(W.smax, s_first);
  for (;;) {
    if constexpr (LDSSTR && TILE::narrow) {        // short-read kernel: one tile, kept between the rounds
#ifdef DH_SPS_LEVEL_V1
      sps_level_block_v1(cons, ref, rcons, rref, m, n, done + 1, S, W.frF, W.frR, W.ndp, T, reachF, reachR, lane);
#else
      sps_level_block(cons, ref, rcons, rref, m, n, done + 1, S, W.frF, W.frR, W.ndp, T, reachF, reachR, lane);
#endif
    } else {
      for (int d = done + 1; d <= S; d += SP_LB)
        sp_level_block2<TILE>(cons, ref, rcons, rref, m, n, d, min(d + SP_LB - 1, S), W.frF, W.frR, W.ndp, T, reachF, reachR, lane);
    }
    __syncthreads();
#ifdef DH_LR_TIMING
    O.t[0] = wall_clock64();
#endif
    done = S;
    // where the tables of levels 0 .. S are read from (see FrTile8 / FrGlobal16)
    constexpr bool TILED8 = LDSSTR && TILE::narrow;
    typedef typename std::conditional<TILED8, FrTile8, FrGlobal16>::type FRV;
    FRV frF, frR;
    if constexpr (TILED8) {
      frF = FrTile8{reinterpret_cast<const uint8_t*>(W.frF), W.ndp, &T.row[0][0][0] + SPS_OFF, &T.row[0][1][0] + SPS_OFF, S};
      frR = FrTile8{reinterpret_cast<const uint8_t*>(W.frR), W.ndp, &T.row[1][0][0] + SPS_OFF, &T.row[1][1][0] + SPS_OFF, S};
    } else {
      frF = FrGlobal16{W.frF, W.ndp};
      frR = FrGlobal16{W.frR, W.ndp};
    }
    // deficit of the unsplit (semi-global) alignment: first level that reaches row m
    int du = -1;
    for (int d0 = 0; d0 <= S && du < 0; d0 += WAVE) {
      const int d = d0 + lane;
      const unsigned long long hit = __ballot(d <= S && reachF[d] >= m);
      if (hit) du = d0 + __builtin_ctzll(hit);
    }
    // With a complete unsplit alignment of deficit du only a split of total deficit < du changes the result (needle.h:152),
    // so the tables are built for the levels 0 .. du - 1 only -- none at all for du = 0, and for a consensus that aligns
    // without a split (the false-positive candidate) usually no level pair is feasible: the cheapest junctions there are.
    const int SE = (du >= 0) ? min(S, du - 1) : S;
    // rows where both sides have cells: forward reaches row r, reverse reaches row m - r
    const int rhi = (SE >= 0) ? min(m, (int)reachF[max(SE, 0)]) : -1, rlo = (SE >= 0) ? max(0, m - (int)reachR[max(SE, 0)]) : 0;
    long long key = 0x7fffffffffffffffll;      // (total deficit << 40) | (row << 20) | column
    int dsel = 0;
    int nlistR = -1;
    // A split of total deficit <= SE needs levels d + e <= SE whose furthest rows meet (reach is non-decreasing in the
    // level): without one the tables cannot resolve the junction and are not built.
    bool feasible = false;
    for (int d0 = 0; SE >= 0 && d0 <= SE && !feasible; d0 += WAVE) {
      const int d = d0 + lane;
      feasible = __ballot(d <= SE && reachF[min(d, SE)] >= 0 && (int)reachR[SE - min(d, SE)] >= m - (int)reachF[min(d, SE)]) != 0ull;
    }
    bool small = false;
    SpsSmall Q;
    if constexpr (LDSSTR && TILE::narrow) {
      if (rlo <= rhi && feasible && SE >= S - 1 && SE <= 8) {   // (the tile holds levels S and S - 1; nine packed levels per diagonal)
        small = sps_lists_small(frF, frR, T.row[0][SE & 1] + SPS_OFF, T.row[1][SE & 1] + SPS_OFF, ND, m, SE, rlo, m - rhi, Q, lane);
        if (small) {
#ifdef DH_LR_TIMING
          O.t[1] = wall_clock64();
#endif
          sps_join_small(Q, m, n, SE, rlo, rhi, key, dsel, lane);
        }
      }
    }
    if (rlo <= rhi && feasible && !small) {
      int nlistF;
      bool tables_done = false;
      if constexpr (LDSSTR && TILE::narrow) {
        if (SE >= S - 1) {   // (the tile holds levels S and S - 1)
          nlistF = sps_deep_list<false>(T.row[0][SE & 1] + SPS_OFF, ND, rlo, W.listF, W.runs_cap, lane);
          nlistR = sps_deep_list<false>(T.row[1][SE & 1] + SPS_OFF, ND, m - rhi, W.listR, W.runs_cap, lane);
        } else {
          nlistF = sp_deep_list(frF, m, n, SE, rlo, W.listF, W.runs_cap, lane);
          nlistR = sp_deep_list(frR, m, n, SE, m - rhi, W.listR, W.runs_cap, lane);
        }
        if (nlistF >= 0 && nlistF <= WAVE && nlistR >= 0 && nlistR <= WAVE) {   // (its first wait covers the list stores too)
          sp_first_columns_both(frF, frR, m, SE, rlo, rhi, W.listF, nlistF, W.cF, m - rhi, m - rlo, W.listR, nlistR, W.cR, lane);
          tables_done = true;
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      } else {
        nlistF = sp_deep_list(frF, m, n, SE, rlo, W.listF, W.runs_cap, lane);
        nlistR = sp_deep_list(frR, m, n, SE, m - rhi, W.listR, W.runs_cap, lane);
      }
      if (!tables_done) {
        if (nlistF >= 0) sp_first_columns_list(frF, m, SE, rlo, rhi, W.listF, nlistF, W.cF, lane);
        else sp_first_columns(frF, m, n, SE, rlo, rhi, W.cF, lane);
        if (nlistR >= 0) sp_first_columns_list(frR, m, SE, m - rhi, m - rlo, W.listR, nlistR, W.cR, lane);
        else sp_first_columns(frR, m, n, SE, m - rhi, m - rlo, W.cR, lane);
      }
#ifdef DH_LR_TIMING
      O.t[1] = wall_clock64();
#endif
      int dbest = 0;
      for (int r0 = rlo; r0 <= rhi; r0 += WAVE) {
        const int r = r0 + lane;
        if (r <= rhi) {
          long long kb = 0x7fffffffffffffffll;
          int db = 0;
          const int32_t* cf = W.cF + r;
          const int32_t* cr = W.cR + (m - r);
          if (SE <= 8) {
            // all table entries of this row in flight at once, then registers only: the minimal e for a given d is the
            // number of levels whose first column is still too far right (first columns shrink with the level).  A wide
            // row range (a consensus that aligns without a split) made the dependent loads below the longest phase.
            int lo_[9], c2_[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
              lo_[q] = (q <= SE) ? sp_ld32(cf + (size_t)q * (m + 1)) : SP_INF;
              c2_[q] = (q <= SE) ? sp_ld32(cr + (size_t)q * (m + 1)) : SP_INF;
            }
            sps_row_best<9>(lo_, c2_, SE, n, r, kb, db);
          } else {
          // two pointers: minimal e for every d (first columns shrink with d, last allowed columns grow with e)
          int e = SE + 1;
          for (int d = 0; d <= SE; ++d) {
            const int lo = sp_ld32(cf + (size_t)d * (m + 1));
            if (lo > n) continue;
            while (e >= 1) {
              const int c2 = sp_ld32(cr + (size_t)(e - 1) * (m + 1));
              if (c2 <= n && lo <= n - c2) --e;
              else break;
            }
            if (e <= SE && d + e <= SE) {
              const long long kk = ((long long)(d + e) << 40) | ((long long)r << 20) | (long long)lo;
              if (kk <= kb) { kb = kk; db = d; }   // ties in d + e: the larger d has the smaller (or equal) column
            }
          }
          }
          if (kb < key) { key = kb; dbest = db; }
        }
      }
      // wave minimum of (key, d)
      long long kmin = key;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const int lo = __shfl_xor((int)(kmin & 0xffffffffll), o), hi = __shfl_xor((int)(kmin >> 32), o);
        const long long w = ((long long)hi << 32) | (unsigned int)lo;
        kmin = w < kmin ? w : kmin;
      }
      const unsigned long long who = __ballot(key == kmin && key != 0x7fffffffffffffffll);
      dsel = who ? __shfl(dbest, __builtin_ctzll(who)) : 0;
      key = kmin;
    }
#ifdef DH_LR_TIMING
    O.t[2] = wall_clock64();
#endif
    const bool have = key != 0x7fffffffffffffffll;
    const int bestD = have ? (int)(key >> 40) : -1;
    if (du >= 0 || have) {
      O.resolved = 1;
      O.levels = S;
      O.unsplit = (du >= 0) ? m - du : SP_UNKNOWN;
      if (!have || (du >= 0 && bestD >= du)) {     // no improving split (needle.h:152)
        O.found = 0;
        O.best = m - du;
        return O;
      }
      O.found = 1;
      O.best = m - bestD;
      O.consLeft = (int)((key >> 20) & 0xfffff);
      O.refLeft = (int)(key & 0xfffff);
      const int cr_ = m - O.consLeft;
      const int eR = bestD - dsel;
      // refRight: last t <= n - refLeft with rev[consRight][t] at deficit eR (needle.h:119-123)
      int rright = 0;
      if (small) {         // (the reverse list and its rows at every level are still in registers)
        const int w = eR >> 2, sh = 8 * (eR & 3);
        const uint32_t pw = (w == 0) ? Q.pR[0] : (w == 1) ? Q.pR[1] : Q.pR[2];
        const int v = (int)((pw >> sh) & 255u) - 1;
        const int t = cr_ + Q.qR - m;
        if (Q.qR >= 0 && t >= 0 && t <= n - O.refLeft && v >= cr_) rright = t;
      } else if (nlistR >= 0) {   // (row consRight >= m - rhi: every diagonal that reaches it is listed)
        for (int i0 = 0; i0 < nlistR; i0 += WAVE) {
          const int i = i0 + lane;
          if (i < nlistR) {
            const int q = sp_ld32(W.listR + i), k = q - m, t = cr_ + k;
            if (t >= 0 && t <= n - O.refLeft && frR.get(eR, q) >= cr_) rright = max(rright, t);
          }
        }
      } else {
        for (int q0 = 0; q0 < ND; q0 += WAVE) {
          const int q = q0 + lane;
          if (q < ND) {
            const int k = q - m, t = cr_ + k;
            if (t >= 0 && t <= n - O.refLeft && frR.get(eR, q) >= cr_) rright = max(rright, t);
          }
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) rright = max(rright, __shfl_xor(rright, o));
      O.refRight = rfl(rright);
#ifdef DH_LR_TIMING
      O.t[3] = wall_clock64();
#endif
      if constexpr (LDSSTR && TILE::narrow) {
        O.runF = O.runR = 0;
        O.nrunsF = sp_trace<FRV, true>(frF, m, n, O.consLeft, O.refLeft, dsel, W.runsF, W.runs_cap, lane, &O.mmF, &O.runF);
        O.nrunsR = sp_trace<FRV, true>(frR, m, n, cr_, O.refRight, eR, W.runsR, W.runs_cap, lane, &O.mmR, &O.runR);
      } else {
        O.nrunsF = sp_trace(frF, m, n, O.consLeft, O.refLeft, dsel, W.runsF, W.runs_cap, lane, &O.mmF);
        O.nrunsR = sp_trace(frR, m, n, cr_, O.refRight, eR, W.runsR, W.runs_cap, lane, &O.mmR);
      }
#ifdef DH_LR_TIMING
      O.t[4] = wall_clock64();
#endif
      if (O.nrunsF < 0 || O.nrunsR < 0) O.resolved = 0;   // run list overflow: dense passes
      return O;
    }
    if (S >= W.smax) return O;       // budget exhausted
    // Not resolved at S: the deficit the junction will need, from how far S levels carried the two sides -- errors are
    // roughly evenly spread, so covering all m consensus rows takes about S * m / (rows covered per side) levels.  Beyond
    // pred_cap the dense strips are cheaper than more levels.
    {
      const long long covered = max(1, (int)reachF[S] + (int)reachR[S]);
      const long long pred = (2ll * S * m) / covered;
      if (pred > W.pred_cap) return O;
    }
    // (even totals dominate: a substitution costs 2.  The short-read kernel goes on in steps of 2 up to 16 -- its slowest
    //  junction is the latency floor of a launch, see tools/n_sweep.py -- the strip kernel in whole level blocks.)
    S = min(W.smax, (S < (LDSSTR ? 16 : 8)) ? S + 2 : (S < 32) ? S * 2 : S + SP_LB);
  }
}

// column masks of the glued alignment (needle.h:196-219) from the two run lists: forward trace reversed, the reference
// gap, the reverse trace in push order.  PL / append as needle_masks (split_main.hpp).  Returns the number of columns.
// run i of a list: from the wavefront's workspace, or from the register of sp_trace<.., true> (lane i holds run i)
struct RunsMem {
  const int32_t* p;
  __device__ __forceinline__ int at(int i) const { return rfl(sp_ld32(p + i)); }
  __device__ __forceinline__ int mine(int i, int n) const { return (i >= 0 && i < n) ? sp_ld32(p + i) : 0; }   // lane-wise, i per lane
};
struct RunsReg {
  int r;
  __device__ __forceinline__ int at(int i) const { return __builtin_amdgcn_readlane(r, i); }
  __device__ __forceinline__ int mine(int i, int n) const { const int v = __shfl(r, max(min(i, WAVE - 1), 0)); return (i >= 0 && i < n) ? v : 0; }
};

template <typename PL, typename APPEND, typename RUNS = RunsMem>
__device__ __forceinline__ int sparse_masks_t(PL& L, RUNS runsF, int nF, RUNS runsR, int nR, int gapref, int maskw,
                                              int lane, int& posC, APPEND append) {
  for (int w = lane; w < maskw; w += WAVE) {
    L.mV[w] = 0;
    L.mR[w] = 0;
    L.mE[w] = 0;
  }
  __syncthreads();
  int pos = 0;
  auto put = [&](int op, int len) {
    const unsigned long long v = (op != 2) ? ~0ull : 0ull, r = (op != 1) ? ~0ull : 0ull;
    for (int k = 0; k < len; k += 64) {
      append(L, pos, min(64, len - k), v, r, lane);
      pos += min(64, len - k);
    }
  };
  for (int i = nF - 1; i >= 0; --i) {
    const int x = runsF.at(i);
    put(x >> 24, x & 0xffffff);
  }
  put(2, gapref);
  posC = pos;
  for (int i = 0; i < nR; ++i) {
    const int x = runsR.at(i);
    put(x >> 24, x & 0xffffff);
  }
  return pos;
}

template <typename PL, typename APPEND>
__device__ __forceinline__ int sparse_masks(PL& L, const int32_t* runsF, int nF, const int32_t* runsR, int nR, int gapref, int maskw,
                                            int lane, int& posC, APPEND append) {
  return sparse_masks_t(L, RunsMem{runsF}, nF, RunsMem{runsR}, nR, gapref, maskw, lane, posC, append);
}

// sparse_masks with every run written by the whole wavefront (one word per lane), the cumulative counts by a prefix sum
// over the lanes, and no letter pass: returns the number of columns, both = columns with a letter in both rows.
// (The equality mask mE is NOT produced: the caller has the match / mismatch counts from the traces.)
template <typename PL, typename RUNS>
__device__ __forceinline__ int sparse_masks_counts_t(PL& L, RUNS runsF, int nF, RUNS runsR, int nR, int gapref, int maskw,
                                                     int lane, int& posC, int& both);

template <typename PL>
__device__ __forceinline__ int sparse_masks_counts(PL& L, const int32_t* runsF, int nF, const int32_t* runsR, int nR, int gapref, int maskw,
                                                   int lane, int& posC, int& both) {
  return sparse_masks_counts_t(L, RunsMem{runsF}, nF, RunsMem{runsR}, nR, gapref, maskw, lane, posC, both);
}

template <typename PL, typename RUNS>
__device__ __forceinline__ int sparse_masks_counts_t(PL& L, RUNS runsF, int nF, RUNS runsR, int nR, int gapref, int maskw,
                                                     int lane, int& posC, int& both) {
  for (int w = lane; w < maskw; w += WAVE) {
    L.mV[w] = 0;
    L.mR[w] = 0;
  }
  __syncthreads();
  int pos = 0;
  auto put = [&](int op, int len) {
    if (len <= 0) return;
    const int w0 = pos >> 6, w1 = (pos + len - 1) >> 6;
    for (int w = w0 + lane; w <= w1; w += WAVE) {
      const int lo = max(pos - w * 64, 0), hi = min(pos + len - w * 64, 64);
      const unsigned long long mk = ((hi >= 64) ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
      if (op != 2) L.mV[w] |= mk;
      if (op != 1) L.mR[w] |= mk;
    }
    pos += len;
  };
  // (run words are independent loads: fetch up to 64 of each list at once)
  const int xF = runsF.mine(nF - 1 - lane, nF);
  const int xR = runsR.mine(lane, nR);
  for (int i = 0; i < nF; ++i) {
    const int x = (i < WAVE) ? __shfl(xF, i) : runsF.at(nF - 1 - i);
    put(x >> 24, x & 0xffffff);
  }
  put(2, gapref);
  posC = pos;
  for (int i = 0; i < nR; ++i) {
    const int x = (i < WAVE) ? __shfl(xR, i) : runsR.at(i);
    put(x >> 24, x & 0xffffff);
  }
  __syncthreads();
  // cumulative letter counts per mask word
  const int nw = (pos + 63) >> 6;
  int b = 0;
  for (int w0 = 0, cv = 0, cr = 0; w0 <= nw; w0 += WAVE) {
    const int w = w0 + lane;
    const unsigned long long mv = (w < nw) ? L.mV[w] : 0ull, mr = (w < nw) ? L.mR[w] : 0ull;
    int pv = __popcll(mv), pr = __popcll(mr);
    b += __popcll(mv & mr);
    int sv = pv, sr = pr;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const int tv = __shfl_up(sv, o), tr = __shfl_up(sr, o);
      if (lane >= o) { sv += tv; sr += tr; }
    }
    if (w <= nw) {
      L.cumV[w] = cv + sv - pv;
      L.cumR[w] = cr + sr - pr;
    }
    cv += __shfl(sv, WAVE - 1);
    cr += __shfl(sr, WAVE - 1);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) b += __shfl_xor(b, o);
  both = b;
  __syncthreads();
  return pos;
}

// sparse_masks_counts_t without LDS: lane w builds word w of the two masks from the run list (a run is a range of columns:
// every lane clips it against its own 64 columns), the letter counts before a word are sums of clipped run lengths, the
// number of columns with both letters is the total length of the 's' runs.  No barrier, no read-modify-write of LDS words,
// no prefix scan; the result feeds split_detect's register path directly.  Alignments of at most MASKREG_COLS columns.
template <typename RUNS>
__device__ __forceinline__ int sparse_masks_regs(RUNS runsF, int nF, RUNS runsR, int nR, int gapref, int lane, int& posC, int& both,
                                                 MaskRegs& M) {
  unsigned long long mv = 0ull, mr = 0ull;
  int cv = 0, cr = 0, pos = 0, b = 0;
  const int base = lane * 64;
  auto put = [&](int op, int len) {   // (op, len, pos: uniform)
    if (len <= 0) return;
    const int lo = min(max(pos - base, 0), 64), hi = min(max(pos + len - base, 0), 64);
    const unsigned long long upto_hi = (hi >= 64) ? ~0ull : ((1ull << (hi & 63)) - 1ull), upto_lo = (lo >= 64) ? ~0ull : ((1ull << (lo & 63)) - 1ull);
    const unsigned long long mk = upto_hi & ~upto_lo;
    const int before = min(max(base - pos, 0), len);   // columns of the run that lie before this lane's word
    if (op != 2) { mv |= mk; cv += before; }
    if (op != 1) { mr |= mk; cr += before; }
    if (op == 0) b += len;
    pos += len;
  };
  const int xF = runsF.mine(nF - 1 - lane, nF);
  const int xR = runsR.mine(lane, nR);
  for (int i = 0; i < nF; ++i) {
    const int x = (i < WAVE) ? __builtin_amdgcn_readlane(xF, i) : runsF.at(nF - 1 - i);
    put(x >> 24, x & 0xffffff);
  }
  put(2, gapref);
  posC = pos;
  for (int i = 0; i < nR; ++i) {
    const int x = (i < WAVE) ? __builtin_amdgcn_readlane(xR, i) : runsR.at(i);
    put(x >> 24, x & 0xffffff);
  }
  both = b;
  M.mv = mv; M.mr = mr; M.cv = cv; M.cr = cr; M.valid = true;
  return pos;
}

}  // namespace dh
