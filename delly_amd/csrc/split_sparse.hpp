// split_sparse.hpp -- short-read alignConsensus (svt != 4) through the SPARSE longNeedle of sparse_needle.hpp: one
// junction per wavefront, furthest-reaching tables instead of the packed dense DP of split_quad.hpp / split_pk.hpp.
//
// A 150 bp consensus against a 1 kb window has 151 x 1001 cells per matrix, but a junction whose consensus carries
// e errors is decided by the cells of deficit <= ~2.5 e: (levels) x (1151 diagonals) table entries.  BASELINE's C2
// consensus sequences (0.5 % substitutions) need 1 .. 9 levels (47 % none, 83 % <= 2, 99.9 % <= 8), so the kernel
// raises the level count per junction (0, 2, 4, 6, 8, 16, 32) and stops as soon as the junction is resolved; what is not
// resolved at 32 levels -- or has letters outside A, C, G, T, N, a consensus beyond 254 bp or a window beyond
// SPS_ND diagonals -- is left to the dense kernels, which skip every junction this kernel finished
// (result.reserved == SPS_DONE).  Everything after the alignment (column masks, _findSplit, _percentIdentity,
// homology, coordinates, alleles) is the shared split_detect stage.
#pragma once
#include "sparse_needle.hpp"
#include "split_main.hpp"

namespace dh {

constexpr int SPS_MMAX = 254;                 // consensus rows the short-read sparse path takes
constexpr int SPS_NMAX = 1280;                // window letters
constexpr int SPS_ND = 1536;                  // diagonals (n + m + 1): one tile, no halo
constexpr int SPS_SMAX = 32;                  // deficit levels before the dense kernels take over
constexpr int SPS_LIST = 1024;                // run / deep-diagonal list capacity
typedef SpTileT<SPS_ND, 0, uint8_t, 2> SpsTile;  // rows <= 254 fit a byte, level d overwrites level d - 2: 6 KB for both matrices

// the junction's strings and the post stage's masks, sized for this kernel's shapes (9.8 KB of LDS per wavefront with
// the tile: 16 wavefronts per CU)
struct __attribute__((aligned(16))) StrLdsS {
  static constexpr bool has_rc = true;
  static constexpr int ref_cap = SPS_NMAX;
  static constexpr int cons_cap = SPS_MMAX + 2;
  uint8_t cons[SPS_MMAX + 2];
  uint8_t rcons[SPS_MMAX + 2];
  uint8_t ref[SPS_NMAX];
  uint8_t rref[SPS_NMAX + 16];   // (+ slack: the compares read 8 letters from any position <= n)
};
struct __attribute__((aligned(16))) PostLdsS {
  unsigned long long mV[MASKW], mR[MASKW], mE[MASKW];
  int32_t cumV[MASKW + 1], cumR[MASKW + 1];
};
struct __attribute__((aligned(16))) SpsLds {
  StrLdsS s;
  union {
    PostLdsS p;
    SpsTile t;
  } u;
  int16_t reachF[SPS_SMAX + 2], reachR[SPS_SMAX + 2];
  int32_t next_item;   // the answer of the work-counter atomic, parked by split_detect's MID hook
};

// bytes of per-wavefront global scratch the kernel needs
__host__ __device__ inline uint64_t sps_scratch_bytes() {
  const uint64_t ndp = (SPS_ND + 2 + 63) & ~63;
  return 4ull * SPS_LIST * 4 + 2ull * (SPS_SMAX + 1) * ndp + 2ull * (SPS_SMAX + 1) * (SPS_MMAX + 1) * 4;
}

// Lane 0 asks the launch's counter for the wavefront's next item.  Written as atomicAdd(counter, 1) on a uniform address the
// compiler turns the atomic into "one lane adds the number of active lanes, every lane derives its value from the result" and
// needs that result AT ONCE (s_waitcnt vmcnt(0) right behind the atomic); and a result that stays live until the next
// junction starts is spilled to scratch memory the moment it is defined -- again a wait for the whole round trip (1 - 3 us
// under load) in front of split_detect instead of under it.  So: the pointer is laundered through vector registers (no
// rewrite), the atomic is issued behind the column masks (in front of them the wait lands in a loop preheader: tools/
// sps_atomic_wait.py), and split_detect's MID hook parks the answer in LDS
// (SpsLds::next_item) a few microseconds later, which is the last use of the register.
__device__ __forceinline__ int sps_ask_next(int32_t* counter, int lane, int keep) {
  uintptr_t p = reinterpret_cast<uintptr_t>(counter);
  uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);
  asm volatile("" : "+v"(plo), "+v"(phi));
  gptr_i32 q = (gptr_i32)(((uintptr_t)phi << 32) | plo);
  int r = keep;
  if (lane == 0) r = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}

// asked: the wavefront's NEXT work index has been asked for and will be in L.next_item (the atomic's round trip runs under the
// masks and split_detect; asked for any earlier -- at the start of the junction -- idle wavefronts at the end of a launch would
// find the last junctions already claimed by busy ones: measured, -15 %)
__device__ __forceinline__ bool process_sparse(const SplitArgs& A, int j, SpsLds& L, uint32_t* scratch, int lane, bool& asked) {
#ifdef DH_LR_TIMING
  const unsigned long long tq0 = wall_clock64();
#endif
  JCtx X;
  const int prior = A.res[j].status;
  junction_setup<KMAX, true, StrLdsS, false, true>(A, j, L.s, X, lane);   // (the host only lists junctions within StrLdsS: no E_LIMIT from here)
  if (!X.go) {   // alignConsensus's early exits (src/split.h:647), unknown svt, limits: the record is final
    const int st = X.out->status;
    const bool final = st == 0;
    if (lane == 0 && final) X.out->reserved = SPS_DONE;
    if (lane == 0 && st != 0 && prior == 0) X.out->status = 0;   // (a limit of THIS kernel's buffers only: the dense kernels start afresh)
    return final;
  }
  const int m = X.m, n = X.n;
  if (m < 1 || n < 1 || m > SPS_MMAX || n > SPS_NMAX || n + m + 1 > SPS_ND) return false;   // dense kernels
  if (X.dirty) return false;   // letters outside A, C, G, T, N (case matters: the forward pass compares raw bytes): dense kernels
  SparseWs W;
  W.ndp = (n + m + 2 + 63) & ~63;
  W.smax = SPS_SMAX;
#ifdef SPS_PRED_CAP
  W.pred_cap = SPS_PRED_CAP;   // (tuning builds: give up when the deficit predicted after a round exceeds this)
#else
  W.pred_cap = 1 << 20;     // (never give up early: an unresolved junction costs a whole dense wavefront of ~1 ms)
#endif
  W.runs_cap = SPS_LIST;
  uint8_t* sp = reinterpret_cast<uint8_t*>(scratch);
  W.runsF = reinterpret_cast<int32_t*>(sp);
  W.runsR = W.runsF + SPS_LIST;
  W.listF = W.runsR + SPS_LIST;
  W.listR = W.listF + SPS_LIST;
  sp += 4ull * SPS_LIST * 4;
  // byte tables (row + 1, 0 = none), level stride ndp: what the tile spills (sparse_needle.hpp, FrTile8)
  W.frF = reinterpret_cast<int16_t*>(sp);
  W.frR = reinterpret_cast<int16_t*>(sp + (size_t)(SPS_SMAX + 1) * W.ndp);
  W.cF = reinterpret_cast<int32_t*>(sp + 2 * (size_t)(SPS_SMAX + 1) * W.ndp);
  W.cR = W.cF + (size_t)(SPS_SMAX + 1) * (m + 1);
  __syncthreads();
#ifdef DH_LR_TIMING
  const unsigned long long tq1 = wall_clock64();
#endif
  const SparseRes sr = sparse_long_needle<SpsTile, true>(L.s.cons, L.s.rcons, L.s.ref, L.s.rref, m, n, W, L.u.t, L.reachF, L.reachR, 0, lane);
  __syncthreads();
#ifdef DH_LR_TIMING
  const unsigned long long tq2 = wall_clock64();
#endif
  if (!sr.resolved) return false;   // dense kernels
  if (lane == 0) {
    X.out->score_unsplit = sr.unsplit;
    X.out->score_best = sr.best;
    X.out->cons_left = sr.found ? sr.consLeft : 0;
    X.out->ref_left = sr.found ? sr.refLeft : 0;
    X.out->ref_right = sr.found ? sr.refRight : n;
  }
  X.consLeft = sr.found ? sr.consLeft : 0;
  X.refLeft = sr.found ? sr.refLeft : 0;
  X.refRight = sr.found ? sr.refRight : 0;
  X.consRight = m - X.consLeft;
  X.go = sr.found != 0;
  int Ltot = 0, posC = 0, pre_ma = -1, pre_mm = -1;
  MaskRegs MR{0ull, 0ull, 0, 0, false};
  if (sr.found) {
    const int gapref = (n - sr.refRight) - sr.refLeft;
    if (A.want_alignment) {   // the alignment rows need every column's letters
      Ltot = sparse_masks_t(L.u.p, RunsReg{sr.runF}, sr.nrunsF, RunsReg{sr.runR}, sr.nrunsR, gapref, MASKW, lane, posC,
                            [](PostLdsS& l, int pos, int cnt, unsigned long long v, unsigned long long r, int ln) { mask_append(l, pos, cnt, v, r, ln); });
      masks_finish(A, X, L.s, L.u.p, Ltot, posC, lane);
    } else {
      int both = 0;
      Ltot = sparse_masks_regs(RunsReg{sr.runF}, sr.nrunsF, RunsReg{sr.runR}, sr.nrunsR, gapref, lane, posC, both, MR);   // (<= 254 + 1 280 + gap columns: one word per lane)
      pre_mm = rfl(sr.mmF + sr.mmR);
      pre_ma = rfl(both) - pre_mm;
    }
  }
  X.uniformize();
  const int pending = sps_ask_next(A.work_counter, lane, 0);   // (behind the masks: in front of them the compiler waits for it at once, tools/sps_atomic_wait.py)
  asked = true;
#ifdef DH_LR_TIMING
  const unsigned long long tq3 = wall_clock64();
#endif
  split_detect(A, X, L.s, L.u.p, X.go, Ltot, posC, lane, pre_ma, pre_mm, true, MR,
               [&]() { if (lane == 0) L.next_item = pending; });
#ifdef DH_LR_TIMING
  if (lane == 0) {   // debug build: phase times in microseconds overwrite diagnostic slots of the record
    const unsigned long long tq4 = wall_clock64();
    X.out->c_start = (int)(tq0 & 0x3fffffffull);                 // start, 10 ns ticks
    X.out->c_end = (int)(tq4 & 0x3fffffffull);                   // end
    X.out->r_start = (int)((sr.t[0] - tq1) / 100);               // levels (+ earlier evaluation rounds), us
    X.out->r_end = (int)((tq2 - sr.t[0]) / 100);                 // last evaluation + traces
    X.out->hom_left = (int)((tq4 - tq2) / 100);                  // masks + split detection
    X.out->hom_right = sr.levels;
#ifdef DH_SPS_FINE
    if (!sr.found) {
      X.out->r_start = (int)((sr.t[1] - sr.t[0]) / 10);
      X.out->r_end = (int)((sr.t[2] - sr.t[1]) / 10);
    }
    if (sr.found) {   // finer: lists + first columns | join + refRight | traces | masks | detect, 100 ns units in five slots
      X.out->r_start = (int)((sr.t[1] - sr.t[0]) / 10);
      X.out->r_end = (int)((sr.t[3] - sr.t[1]) / 10);
      X.out->hom_left = (int)((sr.t[4] - sr.t[3]) / 10);
      X.out->matches = (int)((tq3 - tq2) / 10);
      X.out->mismatches = (int)((tq4 - tq3) / 10);
    }
    // the LAST level block of the junction: before it | clearing the tile | the steps (of its last level) | reductions + end | behind it
    X.out->sr_support = (int)((dh_tl[0] - tq1) / 10);
    X.out->hom_len = (int)((dh_tl[1] - dh_tl[0]) / 10);
    X.out->ci_wiggle = (int)((dh_tl[2] - dh_tl[1]) / 10);
    X.out->cons_bp = (int)((dh_tl[3] - dh_tl[2]) / 10);
    X.out->ins_len = (int)((sr.t[0] - dh_tl[3]) / 10);
#endif
  }
#endif
  if (lane == 0) X.out->reserved = SPS_DONE;
  return true;
}

#ifndef DH_SPARSE_WAVES
#define DH_SPARSE_WAVES 4
#endif
__global__ __launch_bounds__(WAVE, DH_SPARSE_WAVES) void split_sparse_kernel(SplitArgs A) {
  __shared__ SpsLds L;
  const int lane = threadIdx.x;
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  // the first item of every wavefront is its block index -- 4 096 wavefronts asking ONE counter for their first item at launch
  // queue up behind each other for tens of microseconds -- the following ones come from the counter (offset by the grid size)
  bool first = true, asked = false;
  int next_raw = 0;
  for (;;) {
    int w = (int)blockIdx.x;
    if (!first) {
      if (!asked) next_raw = sps_ask_next(A.work_counter, lane, next_raw);   // (a junction that left before its last stage)
      else next_raw = L.next_item;
      w = rfl(next_raw) + (int)gridDim.x;
    }
    first = false;
    asked = false;
    if (w >= A.n_work) break;
    const int j = A.work_list ? A.work_list[w] : w;   // (no list: the batch's junctions in their order)
    // (the lane index is laundered once per junction: address arithmetic on it is then recomputed per junction instead of being
    //  hoisted out of this loop and kept -- or spilled to scratch memory -- for the whole kernel)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (j >= 0 && !process_sparse(A, j, L, scratch, ln, asked) && lane == 0 && A.sps_left) atomicAdd(A.sps_left, 1);
    __syncthreads();
  }
}


// ---- the same for the short-read shapes beyond the byte tile (round 5) ------------------------------------------------------
// split_sparse_kernel stops at a 254 bp consensus (rows in a byte) and 1 280 window letters.  A consensus of 20 reads over an
// insertion-free junction is 200-320 bp, mixed SV types have windows up to 4 |consensus|: in BASELINE's "all SV types" batch one
// junction in seven fell to the packed dense kernels, and a HANDFUL of junctions there costs the step the ~1 ms a dense
// wavefront takes (a latency chain: CHANGELOG.md 0).  This kernel runs the general form of the sparse longNeedle -- int16 rows,
// tiles of 960 diagonals with a halo, tables in the wavefront's HBM workspace: what the strip kernel of lr_kernel.hpp runs,
// with the four strings in LDS -- on every shape the dense kernels take (consensus <= 319, window <= 2 048), one junction
// per wavefront, and what it does not resolve within 32 levels still goes to the dense kernels.
constexpr int SPW_SMAX = 32;
constexpr int SPW_LIST = 2048;
constexpr int SPW_NDP = (NMAX + MMAX + 2 + 63) & ~63;
typedef SpTileT<960, 0, int16_t, 2> SpwTile;

struct __attribute__((aligned(16))) StrLdsW {
  static constexpr bool has_rc = true;
  static constexpr int ref_cap = NMAX;
  static constexpr int cons_cap = MMAX + 9;
  uint8_t cons[MMAX + 9];
  uint8_t rcons[MMAX + 9];
  uint8_t ref[NMAX];
  uint8_t rref[NMAX + 16];   // (+ slack: the compares read 8 letters from any position <= n)
};
struct __attribute__((aligned(16))) SpwLds {
  StrLdsW s;
  union {
    PostLdsS p;
    SpwTile t;
  } u;
  int16_t reachF[SPW_SMAX + 2], reachR[SPW_SMAX + 2];
};

__host__ __device__ inline uint64_t spw_scratch_bytes() {
  return 4ull * SPW_LIST * 4 + 2ull * (SPW_SMAX + 1) * SPW_NDP * 2 + 2ull * (SPW_SMAX + 1) * (MMAX + 1) * 4 + 256;
}
__host__ __device__ inline bool sps_narrow_shape(int m, int n) { return m >= 1 && m <= SPS_MMAX && n <= SPS_NMAX && n + m + 1 <= SPS_ND; }

// 0: finished already, 1: finished here, 2: left to the dense kernels, 3: not finished and not this kernel's shape (a shape
// split_sparse_kernel takes: in a counted launch that kernel has counted it as left; in an uncounted launch the host routed a
// narrow junction here -- its window lengths disagree with the device's -- and it must still reach the dense kernels)
__device__ __forceinline__ int process_sparse_wide(const SplitArgs& A, int j, SpwLds& L, uint32_t* scratch, int lane) {
  if (A.res[j].reserved == SPS_DONE) return 0;
  JCtx X;
  const int prior = A.res[j].status;
  junction_setup<KMAX, true, StrLdsW, false, true>(A, j, L.s, X, lane);
  if (!X.go) {   // alignConsensus's early exits (src/split.h:647), unknown svt, limits: the record is final
    const int st = X.out->status;
    const bool final = st == 0;
    if (lane == 0 && final) X.out->reserved = SPS_DONE;
    if (lane == 0 && st != 0 && prior == 0) X.out->status = 0;   // (a limit of THIS kernel's buffers only: the dense kernels start afresh)
    return final ? 1 : 2;
  }
  const int m = X.m, n = X.n;
  if (m < 1 || n < 1 || m > MMAX || n > NMAX) return 2;
  if (sps_narrow_shape(m, n)) return 3;   // (split_sparse_kernel had it: what it left needs more than 32 levels, or has unclean letters)
  if (X.dirty) return 2;
  SparseWs W;
  W.ndp = (n + m + 2 + 63) & ~63;
  W.smax = SPW_SMAX;
  W.pred_cap = 1 << 20;
  W.runs_cap = SPW_LIST;
  uint8_t* sp = reinterpret_cast<uint8_t*>(scratch);
  W.runsF = reinterpret_cast<int32_t*>(sp);
  W.runsR = W.runsF + SPW_LIST;
  W.listF = W.runsR + SPW_LIST;
  W.listR = W.listF + SPW_LIST;
  sp += 4ull * SPW_LIST * 4;
  W.frF = reinterpret_cast<int16_t*>(sp);
  W.frR = W.frF + (size_t)(SPW_SMAX + 1) * W.ndp;
  W.cF = reinterpret_cast<int32_t*>(W.frR + (size_t)(SPW_SMAX + 1) * W.ndp);
  W.cR = W.cF + (size_t)(SPW_SMAX + 1) * (m + 1);
  __syncthreads();
  const SparseRes sr = sparse_long_needle<SpwTile, true>(L.s.cons, L.s.rcons, L.s.ref, L.s.rref, m, n, W, L.u.t, L.reachF, L.reachR, 8, lane);
  __syncthreads();
  if (!sr.resolved) return 2;
  if (lane == 0) {
    X.out->score_unsplit = sr.unsplit;
    X.out->score_best = sr.best;
    X.out->cons_left = sr.found ? sr.consLeft : 0;
    X.out->ref_left = sr.found ? sr.refLeft : 0;
    X.out->ref_right = sr.found ? sr.refRight : n;
  }
  X.consLeft = sr.found ? sr.consLeft : 0;
  X.refLeft = sr.found ? sr.refLeft : 0;
  X.refRight = sr.found ? sr.refRight : 0;
  X.consRight = m - X.consLeft;
  X.go = sr.found != 0;
  int Ltot = 0, posC = 0, pre_ma = -1, pre_mm = -1;
  if (sr.found) {
    const int gapref = (n - sr.refRight) - sr.refLeft;
    if (A.want_alignment) {
      Ltot = sparse_masks_t(L.u.p, RunsMem{W.runsF}, sr.nrunsF, RunsMem{W.runsR}, sr.nrunsR, gapref, MASKW, lane, posC,
                            [](PostLdsS& l, int pos, int cnt, unsigned long long v, unsigned long long r, int ln) { mask_append(l, pos, cnt, v, r, ln); });
      masks_finish(A, X, L.s, L.u.p, Ltot, posC, lane);
    } else {
      int both = 0;
      Ltot = sparse_masks_counts_t(L.u.p, RunsMem{W.runsF}, sr.nrunsF, RunsMem{W.runsR}, sr.nrunsR, gapref, MASKW, lane, posC, both);
      pre_mm = rfl(sr.mmF + sr.mmR);
      pre_ma = rfl(both) - pre_mm;
    }
  }
  X.uniformize();
  split_detect(A, X, L.s, L.u.p, X.go, Ltot, posC, lane, pre_ma, pre_mm, true);
  if (lane == 0) X.out->reserved = SPS_DONE;
  return 1;
}

// counted != 0: every junction of the list was offered to split_sparse_kernel before, which counted what it left in *sps_left
// (msa() batches: the lengths are not known on the host when the sparse kernels are enqueued) -- a junction finished here is
// taken off that count; counted == 0: the list holds junctions that kernel never saw -- what is left HERE is added.
__global__ __launch_bounds__(WAVE, 3) void split_sparse_wide_kernel(SplitArgs A, uint32_t* wscratch, uint64_t wwords, int counted) {
  __shared__ SpwLds L;
  const int lane = threadIdx.x;
  uint32_t* scratch = wscratch + (size_t)blockIdx.x * wwords;
  for (int w = (int)blockIdx.x; w < A.n_work; w += (int)gridDim.x) {
    const int j = A.work_list[w];
    if (j < 0) continue;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = process_sparse_wide(A, j, L, scratch, ln);
    if (lane == 0 && A.sps_left) {
      if (counted && r == 1) atomicSub(A.sps_left, 1);
      if (!counted && r >= 2) atomicAdd(A.sps_left, 1);   // (r == 3: ADVICE r05 -- never leave a junction unrefined without saying so)
    }
    __syncthreads();
  }
}

}  // namespace dh
