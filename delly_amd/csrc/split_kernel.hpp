// split_kernel.hpp -- gfx950 device code for alignConsensus() (BASELINE unit U):
//   _initBreakpoint/_getSVRef (src/tags.h:151-172, src/split.h:70-163)
//   longNeedle                (src/needle.h:45-222, AlignConfig<true,false>, +1/-1/-1)
//   _findSplit/_percentIdentity/_findHomology/longestHomology/_coordTransform,
//   exact alleles             (src/split.h:166-375,606-637, src/needle.h:13-42)
//
// One junction per 64-lane wavefront.  Lanes own K consecutive DP rows
// ("slots" s = lane*K+i), anti-diagonal skew of one column per lane, row
// hand-off between neighbouring lanes with DPP wave shifts, no LDS traffic in
// the recurrence.  Written for CDNA4 only.
//
// DP domain: V'[s][c] = score[s][c] + s.  With match +1 / mismatch -1 / gap -1
// this turns the vertical move into a plain copy, the diagonal into +2 / +0 and
// makes column 0 identically 0 (see CHANGELOG.md 3.1, "DP formulation").
//
// Pass structure per junction
//   R-pass : reverse-complement DP (rev of needle.h:74-81).  Emits, per cell, a
//            2-bit code of how the row's running maximum (bestRev, :96-103)
//            moved: {below, tie, +1, +2}.  Codes form a per-lane LIFO stack in
//            global scratch (coalesced dwords, 16 cells each).
//   M-pass : forward DP with MIRRORED slots (row r = m - s, lane 63 leads), so
//            that the lane that owns M row r pops exactly the codes it pushed
//            for rev row m - r, in reverse order: bestRev[m-r][n-c] is rebuilt
//            incrementally and the join of needle.h:104-115 is evaluated on
//            the fly.  No score matrix is ever stored.
//   trace  : the two tracebacks (needle.h:154-192) need direction codes only on
//            rows <= consLeft / consRight and columns <= refLeft / refRight;
//            those sub-matrices are recomputed with 2-bit direction output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dellyhip.h"

// DH_SYNC(): the hand-over between the lanes of ONE wavefront (LDS / workspace written by some lanes, read by others).  The
// kernels built from these headers run one wavefront per workgroup, where __syncthreads() is exactly this pair of fences (the
// compiler drops the s_barrier of a single-wave workgroup).  Code that can also run inside a TEAM of wavefronts per workgroup
// (lr_dense_team_kernel: the main wavefront runs the junction, the others only the strips it hands them) must not contain a real
// barrier, so it says DH_SYNC() and the team meets through LDS flags.
#define DH_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

namespace dh {

// LDS reads that are not naturally aligned are served lane by lane on gfx950 (tools/lds_rate.hip, profiles/r04/lds_rate.txt:
// an unaligned ds_read_b64 occupies the LDS pipeline for 29 - 37 cycles, an unaligned ds_read_b32 for 64, their aligned forms
// for 2.5 / 1.6) -- round 3's level loop spent its time there, not in the instruction stream.  Eight letters from any byte
// address are therefore composed from three ALIGNED dwords with two v_alignbyte_b32.
typedef const __attribute__((address_space(3))) uint32_t* sp_lds_u32p;
__device__ __forceinline__ uint64_t sp_lds8a(const uint8_t* p) {
  const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(p);      // (LDS addresses are 32 bits wide)
  sp_lds_u32p q = (sp_lds_u32p)(uintptr_t)(a & ~3u);
  const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, a & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, a & 3u);
  return ((uint64_t)hi << 32) | lo;
}



constexpr int WAVE = 64;
constexpr int NMAX = 2048;            // max |svRefStr| of the short-read kernel
constexpr int KMAX = 5;               // rows per lane: |consensus| <= 64*KMAX-1
constexpr int MMAX = WAVE * KMAX - 1; // 319
constexpr int TRACE_CAP = MMAX + NMAX + 8;
constexpr int MASKW = (MMAX + NMAX + 127) / 64;  // alignment columns / 64
constexpr int NEGBIG = -(1 << 28);
constexpr int NOMATCH = 0x1FF;        // row "character" that equals no byte
constexpr int SCALE_SHIFT = 12;       // M-pass runs on scores << 12 (n < 4096)

struct SplitArgs {
  const dellyhip_junction* junc;
  const uint8_t* cons_base;     // consensus i = cons_base[cons_off[i] .. +cons_len[i])
  const uint64_t* cons_off;
  const int32_t* cons_len;
  const uint8_t* const* chr_seq;  // device table of device pointers
  const int64_t* chr_len;
  int32_t n_chr;
  dellyhip_params p;
  dellyhip_result* res;
  uint8_t* out_blob;            // fixed stride per junction
  uint64_t out_stride;          // [cons MMAX+1][allele ALLELE_CAP][aln 2*TRACE_CAP]
  uint32_t* scratch;            // per resident block
  uint64_t scratch_words;       // words per block
  const uint8_t* ref_base;      // direct mode (single longNeedle): s2 given, no breakpoint logic
  const uint64_t* ref_off;
  const int32_t* ref_len;
  const int32_t* work_list;     // junction indices for this launch (one K bin)
  int32_t n_work;
  int32_t* work_counter;        // zeroed before launch (the atomic of shortpe.h:181)
  int32_t* sps_left;            // junctions split_sparse_kernel left to the dense kernels (nullptr: they run regardless)
  int32_t want_alignment;
  int32_t out_cons_cap, out_allele_cap;  // layout of a junction's out_blob slot: [cons][allele][aln rows]
  int32_t pair_mode;            // 1: split_align_kernel runs behind the packed kernel (deferred junctions only)
};

constexpr int DH_DEFERRED = 1;  // transient result.status: not representable in the packed kernel
constexpr int SPS_DONE = 2;     // transient result.reserved: finished by split_sparse_kernel (the dense kernels skip the junction)

constexpr int OUT_CONS_CAP = MMAX + 1;
constexpr int OUT_ALLELE_CAP = NMAX + MMAX + 8;
constexpr int OUT_ALN_CAP = 2 * TRACE_CAP;

// ---- small device helpers -------------------------------------------------

// A pointer that was LOADED from memory (the chromosome table) or that arrives as the argument of a called function has no
// address space the compiler can see: every access through it becomes a flat_* instruction, which counts against the LDS
// counter as well as the memory counter (LDS reads wait for outstanding global loads) and may alias LDS (loads cannot pass
// LDS stores).  Accesses made THROUGH a pointer of these types are global_* instructions.  (A round trip
// generic -> global -> generic does not help: the pair of casts is folded away before address spaces are inferred.)
typedef const __attribute__((address_space(1))) uint8_t* gptr_cu8;
typedef __attribute__((address_space(1))) uint32_t* gptr_u32;
typedef __attribute__((address_space(1))) int32_t* gptr_i32;

__device__ __forceinline__ int dpp_from_prev(int src, int old) {  // lane l <- lane l-1
  return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_next(int src, int old) {  // lane l <- lane l+1
  return __builtin_amdgcn_update_dpp(old, src, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rfl64u(uint64_t v) {
  return ((uint64_t)(uint32_t)rfl((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)rfl((int)(uint32_t)(v & 0xffffffffull));
}
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t ld_scratch(const uint32_t* p) {
  // L1-bypassing load: scratch words are re-written by this wave for every junction
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same without branches (a nibble table over 'A' .. 'Z' + 6): written as a chain of comparisons the compiler turns
// letter_code() into a tree of divergent branches -- ~60 scalar instructions per letter in the LCS loop, which took 245 of a
// junction's 1 750 us (round 5, tools/msa_phases.py)
__device__ __forceinline__ int letter_code_bf(uint8_t c) {
  const uint32_t idx = (uint32_t)c - (uint32_t)'A';
  const unsigned long long tbl = (idx & 16u) ? 0xffffffffffff3fffull : 0xff4ffffff2fff1f0ull;
  const int v = (int)((tbl >> (4u * (idx & 15u))) & 15ull);
  return (idx < 32u && v != 15) ? v : -1;
}

__device__ __forceinline__ uint8_t upc(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
// complement of an (already upper-cased) base; 0 when outside ACGTN
// (without branches: a nibble table over 'A' .. 'A' + 31 says whether the byte is one of A, C, G, T, N -- 0 .. 4 -- and A <-> T is
//  x ^ 0x15, C <-> G is x ^ 0x04, bit 1 of the byte tells the two pairs apart; the chain of comparisons this replaces compiled into a
//  tree of divergent branches per letter, round 5)
__device__ __forceinline__ uint8_t comp_acgtn(uint8_t u) {
  const uint32_t idx = (uint32_t)u - (uint32_t)'A';
  const unsigned long long tbl = (idx & 16u) ? 0xffffffffffff3fffull : 0xff4ffffff2fff1f0ull;
  const uint32_t v = (uint32_t)((tbl >> (4u * (idx & 15u))) & 15ull);
  const bool ok = idx < 32u && v != 15u;
  const uint8_t sw = (uint8_t)(u ^ ((u & 2u) ? 0x04u : 0x15u));
  return ok ? ((v == 4u) ? (uint8_t)'N' : sw) : (uint8_t)0;
}
// byte i of reverseComplement(str) (util.h:549-563): complement of the upper-cased mirrored
// byte, or the ORIGINAL byte i when that is not one of ACGTN
__device__ __forceinline__ uint8_t rc_at(const uint8_t* str, int len, int i) {
  uint8_t r = comp_acgtn(upc(str[len - 1 - i]));
  return r ? r : str[i];
}
// the output switch of needle.h:209-217 (adds '-' -> '-', everything else -> 0)
__device__ __forceinline__ uint8_t outmap(uint8_t ch) {
  return ch == '-' ? '-' : comp_acgtn(ch);
}

// forward strings only (packed DP kernel: reverse complements are derived on the fly)
struct __attribute__((aligned(16))) StrLdsFwd {
  static constexpr bool has_rc = false;
  static constexpr int ref_cap = NMAX;
  static constexpr int cons_cap = MMAX + 1;
  uint8_t cons[MMAX + 1];   // s1
  uint8_t ref[NMAX];        // s2 = svRefStr
};
// strings of one junction (LDS)
struct __attribute__((aligned(16))) StrLds {
  static constexpr bool has_rc = true;
  static constexpr int ref_cap = NMAX;
  static constexpr int cons_cap = MMAX + 1;
  uint8_t cons[MMAX + 1];   // s1
  uint8_t rcons[MMAX + 1];  // reverseComplement(s1), util.h:549-563 semantics
  uint8_t ref[NMAX];        // s2 = svRefStr
  uint8_t rref[NMAX];       // reverseComplement(s2)
};
// scratch of the post-processing stage (one junction at a time, LDS)
struct __attribute__((aligned(16))) PostLds {
  uint8_t trF[TRACE_CAP];   // forward traceback ops in push order (0 's',1 'v',2 'h')
  uint8_t trR[TRACE_CAP];
  unsigned long long mV[MASKW], mR[MASKW], mE[MASKW];  // column masks: var/ref present, equal
  int32_t cumV[MASKW + 1], cumR[MASKW + 1];
};

// reference-window segment (piece of _getSVRef's concatenation)
struct Seg {
  const uint8_t* base;  // chromosome pointer
  int32_t beg, len;     // [beg, beg+len)
  int32_t rc;           // split.h:78-91 style reverse complement
};

// ---- DP passes --------------------------------------------------------------

// R-pass.  rows: slot s (valid 1..m) holds rcons[s-1]; columns: rref[c-1].
// Pushes (steps+15)/16 blocks of K*64 dwords.  Returns per-slot final V' and
// running max.
template <int K>
__device__ __forceinline__ void pass_R(const StrLds& L, int m, int n, uint32_t* stack, int lane,
                                       int (&hfin)[K], int (&brfin)[K]) {
  int a[K], hg[K], h[K], br[K];
  uint32_t acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    a[i] = (s >= 1 && s <= m) ? (int)L.rcons[s - 1] : NOMATCH;
    hg[i] = (s >= 1 && s < m) ? -1 : 0;
    h[i] = 0;
    br[i] = 0;
    acc[i] = 0;
  }
  const int T = n + 63;
  const int nblk = (T + 15) >> 4;
  int upPrev = NEGBIG;
  int b = NOMATCH;
  int c = -lane;  // column of this lane at step t is t - lane; incremented before use
  for (int blk = 0; blk < nblk; ++blk) {
    // characters for lane 0 of the next 16 steps: rref[blk*16 + f], f = 0..15
    int ci = blk * 16 + (lane & 15);
    int chunk = (ci < n) ? (int)L.rref[ci] : NOMATCH;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      int newc = __builtin_amdgcn_readlane(chunk, f);
      b = dpp_from_prev(b, newc);
      int recv = dpp_from_prev(h[K - 1], NEGBIG);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)n) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          int x = diag + ((a[i] == b) ? 2 : 0);
          int z = h[i] + hg[i];
          int nv = max3i(x, up, z);
          diag = h[i];
          up = nv;
          h[i] = nv;
          int d = nv - br[i];
          br[i] = max(br[i], nv);
          int dm = max(d, -1);
          acc[i] = acc[i] + ((uint32_t)dm << (2 * f));
        }
      }
      upPrev = recv;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      stack[((size_t)blk * K + i) * WAVE + lane] = acc[i] + 0x55555555u;
      acc[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    hfin[i] = h[i];
    brfin[i] = br[i];
  }
}

// M-pass with join.  Mirrored slots: slot s is M row r = m - s; lane 63 leads.
// brfin[] = final running maxima of the R-pass (same slots).  Outputs: the
// per-slot best join key ((sum' << 12) | (4095 - c)) and the final V' of row m.
template <int K>
__device__ __forceinline__ void pass_M(const StrLds& L, int m, int n, const uint32_t* stack, int lane,
                                       const int (&brfin)[K], int (&bestkey)[K], int& hrow_m) {
  int a[K], hg[K], h[K], bm[K], g[K];
  uint32_t dw[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    int r = m - s;
    a[i] = (r >= 1) ? (int)L.cons[r - 1] : NOMATCH;
    hg[i] = (r >= 1 && r < m) ? -(1 << SCALE_SHIFT) : 0;
    h[i] = 0;
    bm[i] = 0;
    g[i] = (r >= 0) ? (brfin[i] << SCALE_SHIFT) : NEGBIG;
    bestkey[i] = (r >= 0) ? (bm[i] + g[i] + 4095) : (int)0x80000000;  // column 0 candidate
    dw[i] = 0;
  }
  const int T = n + 63;
  const int nblk = (T + 15) >> 4;
  int upPrev = NEGBIG;
  int b = NOMATCH;
  // consumer step t = T - t' + 1, t' descending from nblk*16; column c = t - 63 + lane
  int c = (T - nblk * 16) - 63 + lane;  // value at "t'= nblk*16 + 1"; incremented before use
  for (int blk = nblk - 1; blk >= 0; --blk) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      uint32_t w = ld_scratch(&stack[((size_t)blk * K + i) * WAVE + lane]);
      // codes {0 below,1 tie,2 +1,3 +2} -> delta fields {0,0,1,2}
      uint32_t hi = (w >> 1) & 0x55555555u, lo = w & 0x55555555u;
      dw[i] = (hi & ~lo) | ((hi & lo) << 1);
    }
    // lane 63's characters for the 16 steps of this block: step f (15..0) has
    // t = T - (blk*16+f+1) + 1, column t, character ref[t-1] = ref[T - blk*16 - f - 1]
    int ci = T - blk * 16 - 16 + (lane & 15);  // f = 15 - (lane&15)
    int chunk = (ci >= 0 && ci < n) ? (int)L.ref[ci] : NOMATCH;
#pragma unroll
    for (int f = 15; f >= 0; --f) {
      int newc = __builtin_amdgcn_readlane(chunk, 15 - f);
      b = dpp_from_next(b, newc);
      int recv = dpp_from_next(h[0], NEGBIG);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)n) {
        int cinv = 4095 - c;
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
          int x = diag + ((a[i] == b) ? (2 << SCALE_SHIFT) : 0);
          int z = h[i] + hg[i];
          int nv = max3i(x, up, z);
          diag = h[i];
          up = nv;
          h[i] = nv;
          bm[i] = max(bm[i], nv);
          int delta = (int)((dw[i] >> (2 * f)) & 3u);
          g[i] = g[i] - (delta << SCALE_SHIFT);
          bestkey[i] = max(bestkey[i], bm[i] + g[i] + cinv);
        }
      }
      upPrev = recv;
    }
  }
  hrow_m = h[0];  // meaningful in lane 0 (slot 0 = row m)
}

// Direction pass: recomputes rows 0..rmax x columns 1..ncols of a needle
// matrix (natural slots: slot s = row s) and stores 2-bit direction codes
// (1 = vertical first, 2 = horizontal, 0 = diagonal: needle.h:160-170).
template <int K>
__device__ __forceinline__ void pass_dir(const uint8_t* rowstr, const uint8_t* colstr, int m, int rmax, int ncols,
                                         uint32_t* dirs, int lane) {
  int a[K], hg[K], h[K];
  uint32_t acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    a[i] = (s >= 1 && s <= m) ? (int)rowstr[s - 1] : NOMATCH;
    hg[i] = (s >= 1 && s < m) ? -1 : 0;
    h[i] = 0;
    acc[i] = 0;
  }
  const int T = ncols + rmax / K;
  const int nblk = (T + 15) >> 4;
  int upPrev = NEGBIG;
  int b = NOMATCH;
  int c = -lane;
  for (int blk = 0; blk < nblk; ++blk) {
    int ci = blk * 16 + (lane & 15);
    int chunk = (ci < ncols) ? (int)colstr[ci] : NOMATCH;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      int newc = __builtin_amdgcn_readlane(chunk, f);
      b = dpp_from_prev(b, newc);
      int recv = dpp_from_prev(h[K - 1], NEGBIG);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)ncols) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          int x = diag + ((a[i] == b) ? 2 : 0);
          int z = h[i] + hg[i];
          int nv = max3i(x, up, z);
          uint32_t code = (nv == up) ? 1u : ((nv == z) ? 2u : 0u);
          diag = h[i];
          up = nv;
          h[i] = nv;
          acc[i] |= code << (2 * f);
        }
      }
      upPrev = recv;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      dirs[((size_t)blk * K + i) * WAVE + lane] = acc[i];
      acc[i] = 0;
    }
  }
}

// Traceback over stored 2-bit codes (uniform across the wave).  A dependent L2 round trip per
// step would dominate the post-processing stage, so the codes are read in WINDOWS: lane l
// fetches the code word of cell (rr-l, cc-l) -- the diagonal through the current cell -- and the
// walk continues out of registers for as long as the path stays inside the fetched words (16
// columns per row).  Inside a window, diagonal RUNS are taken in one step: every lane decodes
// the cell of its own row on the current diagonal, a ballot gives the length of the run of
// diagonal codes, the run's ops are stored lane-parallel.  Only gap ops advance one at a time.
//
// Geo::locate(r, c, word_index, t): code word of cell (r, c) and its step index t (field
// t & 15 of word t >> 4).  ED = false: needle codes 0 diag 's', 1 vertical 'v' (row only),
// 2 horizontal 'h'.  ED = true: EDLIB_EDOP codes 0 MATCH / 3 MISMATCH diagonal, 1 INSERT
// (column only), 2 DELETE (row only).  Pushes the ops to tr[] until a border is reached and
// returns their number; rr / cc are left at the border cell.
template <bool ED, typename Geo>
__device__ __forceinline__ int traceback_runs(const Geo& G, int& rr, int& cc, uint8_t* tr, int lane) {
  constexpr uint32_t ROWCODE = ED ? 2u : 1u;
  int tl = 0;
  rr = rfl(rr);
  cc = rfl(cc);
  while (rr > 0 && cc > 0) {
    const int r = rr - lane, c = cc - lane;
    uint32_t w = 0;
    int tw = -1;
    if (r >= 1 && c >= 1) {
      size_t wi;
      int t;
      G.locate(r, c, wi, t);
      tw = t >> 4;
      w = ld_scratch(G.base + wi);
    }
    int l = 0;   // rr == (rr at fetch time) - l: lane x holds the word of row rr - (x - l)
    bool inwin = true;
    while (inwin) {
      const int d = lane - l;
      const int cx = cc - d;
      bool valid = (d >= 0) && (r >= 1) && (cx >= 1);
      int tx = 0;
      if (valid) {
        size_t wi;
        G.locate(r, cx, wi, tx);
        valid = ((tx >> 4) == tw);
      }
      const uint32_t code = valid ? ((w >> (2 * (tx & 15))) & 3u) : 4u;
      const bool isd = ED ? (code == 0u || code == 3u) : (code == 0u);
      const unsigned long long dm = __ballot(isd) >> l;
      const int L = (~dm == 0ull) ? WAVE : __builtin_ctzll(~dm);
      if (L > 0) {
        if (d >= 0 && d < L) tr[tl + d] = (uint8_t)code;
        tl += L;
        rr -= L;
        cc -= L;
        l += L;
      }
      if (rr <= 0 || cc <= 0 || l >= WAVE) {
        inwin = false;
      } else {
        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)code, l);
        if (cl == 4u) {
          inwin = false;   // the path left the fetched word of this row
        } else {
          if (lane == 0) tr[tl] = (uint8_t)cl;
          ++tl;
          if (cl == ROWCODE) { --rr; ++l; }
          else --cc;
          if (rr <= 0 || cc <= 0 || l >= WAVE) inwin = false;
        }
      }
    }
  }
  return tl;
}

// code-word geometry of pass_dir<K> / pass_ed<K, true>: word ((t >> 4)*K + i)*64 + lane(r), t = c + lane(r) - 1
template <int K>
struct GeoK {
  const uint32_t* base;
  __device__ __forceinline__ void locate(int r, int c, size_t& wi, int& t) const {
    const int lo = r / K, i = r - lo * K;
    t = c + lo - 1;
    wi = ((size_t)(t >> 4) * K + i) * WAVE + lo;
  }
};

// needle traceback (needle.h:154-192): ops 0 's', 1 'v', 2 'h'; the remaining straight run is
// returned as tailV / tailH
template <int K>
__device__ __forceinline__ int traceback(const uint32_t* dirs, int rr, int cc, uint8_t* tr, int lane, int& tailV,
                                         int& tailH) {
  GeoK<K> G{dirs};
  const int tl = traceback_runs<false>(G, rr, cc, tr, lane);
  tailV = rr;  // column 0: only vertical moves remain
  tailH = cc;  // row 0: only horizontal moves remain
  return tl;
}

// ---- column-mask stream -----------------------------------------------------

// appends `cnt` (<=64) bits of (v, r) at bit position pos of the LDS masks
template <typename PL>
__device__ __forceinline__ void mask_append(PL& L, int pos, int cnt, unsigned long long v, unsigned long long r,
                                            int lane) {
  if (cnt <= 0) return;
  unsigned long long keep = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);
  v &= keep;
  r &= keep;
  int w = pos >> 6, o = pos & 63;
  if (lane == 0) {
    L.mV[w] |= v << o;
    L.mR[w] |= r << o;
    if (o && (o + cnt > 64)) {
      L.mV[w + 1] |= v >> (64 - o);
      L.mR[w + 1] |= r >> (64 - o);
    }
  }
}

__device__ __forceinline__ int cnt_before(const unsigned long long* mk, const int32_t* cum, int pos) {
  int w = pos >> 6, o = pos & 63;
  unsigned long long x = mk[w] & ((o == 0) ? 0ull : (~0ull >> (64 - o)));
  return cum[w] + __popcll(x);
}
// next position >= pos (< total) whose bit in (mk ^ inv) is set; returns total if none
__device__ __forceinline__ int next_set(const unsigned long long* mk, unsigned long long inv, int pos, int total) {
  while (pos < total) {
    int w = pos >> 6, o = pos & 63;
    unsigned long long x = (mk[w] ^ inv) >> o;
    if (x) {
      int p = pos + __builtin_ctzll(x);
      return p < total ? p : total;
    }
    pos = (w + 1) << 6;
  }
  return total;
}
// position of the k-th (1-based) set bit of mk, or total if fewer
__device__ __forceinline__ int select_bit(const unsigned long long* mk, const int32_t* cum, int k, int total) {
  int nw = (total + 63) >> 6;
  for (int w = 0; w < nw; ++w) {
    if (cum[w + 1] >= k) {
      unsigned long long x = mk[w];
      int need = k - cum[w];
      for (int q = 1; q < need; ++q) x &= x - 1;
      return (w << 6) + __builtin_ctzll(x);
    }
  }
  return total;
}

// The same three queries with the masks in REGISTERS: lane w holds word w of the two masks and the letter counts before it
// (alignments of fewer than 64 x 64 columns -- every short-read shape).  _findSplit on the LDS masks is a chain of dependent LDS
// round trips (one per mask word it walks over: a 700-column deletion is eleven of them, ~130 cycles each); here a query is a
// ballot and two v_readlane.  All arguments and results are wavefront-uniform.
struct MaskRegs {
  unsigned long long mv, mr;   // word `lane` of the var / ref masks (0 beyond the alignment)
  int cv, cr;                  // var / ref letters in the columns before 64 * lane
  bool valid;
};
constexpr int MASKREG_COLS = WAVE * 64 - 1;   // (position `total` itself must still name a lane)
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v & 0xffffffffull), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int cnt_before_reg(unsigned long long mk, int cum, int pos) {
  const int w = pos >> 6, o = pos & 63;
  const unsigned long long x = readlane64(mk, w) & ((o == 0) ? 0ull : (~0ull >> (64 - o)));
  return __builtin_amdgcn_readlane(cum, w) + __popcll(x);
}
// next position >= pos (< total) whose bit in x is set (x: one word per lane, no bits at or beyond total); total if none
__device__ __forceinline__ int next_set_reg(unsigned long long x, int pos, int total, int lane) {
  if (pos >= total) return total;
  const int w = pos >> 6, o = pos & 63;
  const unsigned long long y = (lane > w) ? x : (lane == w) ? (x & (~0ull << o)) : 0ull;
  const unsigned long long b = __ballot(y != 0ull);
  if (!b) return total;
  const int l = __builtin_ctzll(b);
  return (l << 6) + __builtin_ctzll(readlane64(y, l));
}
// position of the k-th (1-based) set bit, or total if fewer
__device__ __forceinline__ int select_bit_reg(unsigned long long mk, int cum, int k, int total, int lane) {
  const int pc = __popcll(mk);
  const unsigned long long b = __ballot(k >= 1 && cum < k && k <= cum + pc);
  if (!b) return total;
  const int w = __builtin_ctzll(b);
  unsigned long long x = readlane64(mk, w);
  const int need = k - __builtin_amdgcn_readlane(cum, w);
  for (int q = 1; q < need; ++q) x &= x - 1;
  return (w << 6) + __builtin_ctzll(x);
}

// longestHomology(s1, s2, -1)  src/needle.h:13-42 (band k = 1) on strided views
// a[i] = A[ia + i*da], b[j] = B[ib + j*db]; m, n lengths.  Uniform serial code.
// getA(i) = a[i], getB(j) = b[j]; rows 1 .. min(m, rows) are run; finished = the function's value is final
template <typename GA, typename GB>
__device__ __forceinline__ int longest_homology_rows(GA getA, GB getB, int m, int n, int rows, bool& finished) {
  // rolling band: prev row values at columns row-2..row (relative), cur row
  // mat[row][col] valid for |row-col| <= 1; everything else is never read.
  // prev[] indexed by h+1 (h = col-row in -1..1)
  int pm1 = 0, p0 = 0, pp1 = 0;  // row-1: cols row-2, row-1, row   (h = -1,0,1 of row-1)
  // row 0: mat[0][0] = 0, mat[0][1] = -1
  p0 = 0;     // mat[0][0]   (h=0 of row 0)
  pp1 = -1;   // mat[0][1]   (h=1 of row 0)
  pm1 = 0;    // unused for row 0
  finished = true;
  for (int row = 1; row <= min(m, rows); ++row) {
    int best = -2;
    int cm1 = 0, c0 = 0, cp1 = 0;
    bool vm1 = false, v0 = false;
    int ach = getA(row - 1);
    // h = -1 : col = row-1
    {
      int col = row - 1;
      if (col >= 1 && col <= n) {
        // diag = mat[row-1][col-1] = (row-1, h=-1) -> pm1 ; for row==1,col==0 skipped
        int v = pm1 + ((ach == getB(col - 1)) ? 0 : -1);
        // vertical: mat[row-1][col]: row-1-col = 0 in band -> p0
        v = max(v, p0 - 1);
        // horizontal: row-col+1 = 2 > k : not allowed
        cm1 = v;
        vm1 = true;
        if (v > best) best = v;
      } else if (col == 0 && row == 1) {
        cm1 = -1;  // mat[1][0] initialised by needle.h:25
        vm1 = true;
      }
    }
    // h = 0 : col = row
    {
      int col = row;
      if (col >= 1 && col <= n) {
        int v = p0 + ((ach == getB(col - 1)) ? 0 : -1);
        v = max(v, pp1 - 1);                 // vertical: row-1-col = -1 in band
        if (vm1) v = max(v, cm1 - 1);        // horizontal: row-col+1 = 1 in band
        else v = max(v, 0 - 1);              // mat[row][col-1] never written: value-initialised 0
        c0 = v;
        v0 = true;
        if (v > best) best = v;
      }
    }
    // h = +1 : col = row+1
    {
      int col = row + 1;
      if (col >= 1 && col <= n) {
        int v = pp1 + ((ach == getB(col - 1)) ? 0 : -1);
        // vertical: row-1-col = -2 : not allowed
        if (v0) v = max(v, c0 - 1);  // horizontal: row-col+1 = 0 in band
        else v = max(v, 0 - 1);
        cp1 = v;
        if (v > best) best = v;
      }
    }
    if (best < -1) return row - 1;
    pm1 = cm1;
    p0 = c0;
    pp1 = cp1;
  }
  finished = rows >= m;
  return 0;
}
// The usual junction has 0 - 3 bp of homology: the first seven rows only touch a[0 .. 6] and b[0 .. 7], which are fetched as two
// 8-letter words (one LDS round trip instead of one per row) and walked with scalar code; longer homologies start over letter by
// letter.  (Strings in LDS; a descending view needs its 8 letters to start inside the array.)
template <bool LDS8 = false>   // LDS8: both strings live in LDS (sp_lds8a reads through an LDS address)
__device__ __forceinline__ int longest_homology(const uint8_t* A, int ia, int da, int m, const uint8_t* B, int ib,
                                                int db, int n) {
  bool finished = false;
  if (LDS8 && (da == 1 || ia >= 7) && (db == 1 || ib >= 7) && ia >= 0 && ib >= 0) {
    const uint64_t pa = rfl64u((da == 1) ? sp_lds8a(A + ia) : __builtin_bswap64(sp_lds8a(A + ia - 7)));
    const uint64_t pb = rfl64u((db == 1) ? sp_lds8a(B + ib) : __builtin_bswap64(sp_lds8a(B + ib - 7)));
    const int h = longest_homology_rows([&](int i) { return (int)((pa >> (8 * i)) & 255u); }, [&](int j) { return (int)((pb >> (8 * j)) & 255u); },
                                        m, n, 7, finished);
    if (finished) return h;
  }
  return longest_homology_rows([&](int i) { return (int)A[ia + i * da]; }, [&](int j) { return (int)B[ib + j * db]; }, m, n, m, finished);
}

}  // namespace dh
