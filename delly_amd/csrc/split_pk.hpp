// split_pk.hpp -- packed longNeedle passes: TWO junctions per 64-lane wavefront.
//
// rocprof (profiles/r01) shows the one-junction kernel at ~100 % VALU issue with
// ~4 cycles per wave64 integer op, so the only lever left is instructions per cell.
// All DP quantities of the short-read shapes fit 16 bits (V' <= 2m <= 638), so the two
// halves of every 32-bit register carry the same cell of two different junctions
// (lo = junction A, hi = junction B) and every recurrence op becomes one v_pk_*_i16:
//   * match score without compares: column bytes are one-hot coded (bit idx of the byte in
//     the junction's row alphabet, <= 15 distinct bytes), rows keep a shift; (b >> sh) & 2 is
//     the +2 / +0 diagonal bonus of the V' domain for both halves at once;
//   * the join keeps a 16-bit best sum per row and the column of its last strict improvement
//     (mask = (best - newbest) >> 15 arithmetic, v_bfi), instead of the 32-bit (sum<<12|col) key;
//   * the 2-bit running-max codes of both junctions share one dword (8 steps x 2 halves).
// Junction B may have a shorter reference (nB <= nA): in the R-pass its columns beyond nB are
// all-zero one-hot codes (they match nothing and only extend the code stack), in the M-pass it
// starts nA-nB columns late; in the V' domain a never-matching column reproduces column 0, so
// B's state is untouched until its first real column (derivation in CHANGELOG.md 3.4).
#pragma once
#include "split_main.hpp"

namespace dh {

typedef uint32_t pk;
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2v as_s(pk x) { return __builtin_bit_cast(s2v, x); }
__device__ __forceinline__ u2v as_u(pk x) { return __builtin_bit_cast(u2v, x); }
__device__ __forceinline__ pk from_s(s2v x) { return __builtin_bit_cast(pk, x); }
__device__ __forceinline__ pk from_u(u2v x) { return __builtin_bit_cast(pk, x); }
__device__ __forceinline__ pk pk_add(pk a, pk b) { return from_s(as_s(a) + as_s(b)); }
__device__ __forceinline__ pk pk_sub(pk a, pk b) { return from_s(as_s(a) - as_s(b)); }
__device__ __forceinline__ pk pk_max(pk a, pk b) { return from_s(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ pk pk_shr(pk a, pk sh) { return from_u(as_u(a) >> as_u(sh)); }        // logical, per-half amounts
__device__ __forceinline__ pk pk_shl_c(pk a, int sh) { return from_u(as_u(a) << (unsigned short)sh); }
// sign mask of both halves.  Inline asm: written as a C shift hipcc rewrites it into
// v_cmp_lt_i16_sdwa + v_cndmask + v_perm (5 VALU instead of 1).  Output only feeds plain VALU.
__device__ __forceinline__ pk pk_sar15(pk a) {
#ifndef DH_NO_ASMSAR
  pk d;
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a));
  return d;
#else
  return from_s(as_s(a) >> (short)15);
#endif
}
// lane l <- lane l-1 / l+1 with ZERO fill at the wave edge (no v_mov to seed the destination).
// A zero from outside the wave is harmless: the edge slot is row 0 / a padding row whose row
// shift (15) never matches, so it stays 0 in the V' domain.
#ifndef DH_NO_ZFILL
__device__ __forceinline__ pk dppz_from_prev(pk src) {
  return (pk)__builtin_amdgcn_update_dpp(0, (int)src, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ pk dppz_from_next(pk src) {
  return (pk)__builtin_amdgcn_update_dpp(0, (int)src, 0x130, 0xf, 0xf, true);
}
#else
__device__ __forceinline__ pk dppz_from_prev(pk src) { return (pk)dpp_from_prev((int)src, 0); }
__device__ __forceinline__ pk dppz_from_next(pk src) { return (pk)dpp_from_next((int)src, 0); }
#endif
__device__ __forceinline__ int lo16(pk x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int hi16(pk x) { return (int)(short)(x >> 16); }
__device__ __forceinline__ pk mk(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

constexpr pk NEG2 = 0xC000C000u;   // -16384 in both halves
constexpr int NEG16 = -16000;

struct __attribute__((aligned(16))) PairLds {
  StrLdsFwd s[2];
  uint8_t tab[2][2][256];  // [junction][0: alphabet of cons, 1: alphabet of rcons] byte -> index 1..15 (0: absent)
};

// byte -> index table of the distinct bytes of rowstr; returns their number
template <bool RC>
__device__ __forceinline__ int build_table(const uint8_t* rowstr, int len, uint8_t* tab, int lane) {
  for (int v = lane; v < 256; v += WAVE) tab[v] = 0;
  __syncthreads();
  for (int i = lane; i < len; i += WAVE) tab[RC ? rc_at(rowstr, len, i) : rowstr[i]] = 1;
  __syncthreads();
  int base = 0;
  for (int chunk = 0; chunk < 4; ++chunk) {
    int v = chunk * 64 + lane;
    bool pres = tab[v] != 0;
    unsigned long long bm = __ballot(pres);
    unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int idx = base + __popcll(bm & below) + 1;
    if (pres) tab[v] = (uint8_t)min(idx, 255);
    base += __popcll(bm);
  }
  __syncthreads();
  return base;
}

__device__ __forceinline__ uint32_t onehot(const uint8_t* tab, uint8_t ch) {
  uint32_t idx = tab[ch];
  return idx ? (1u << idx) : 0u;
}

// packed R-pass (both junctions' reverse-complement DP).  Pushes 2 dwords per 16 steps per slot.
template <int K>
__device__ __forceinline__ void pass_R2(const StrLdsFwd& SA, const StrLdsFwd& SB, const uint8_t* tabA, const uint8_t* tabB,
                                        int mA, int nA, int mB, int nB, uint32_t* stack, int lane, pk (&hfin)[K],
                                        pk (&hsnap)[K], pk (&brfin)[K]) {
  pk rsh[K], hg[K], h[K], br[K], accA[K], accB[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    int shA = (s >= 1 && s <= mA) ? (int)tabA[rc_at(SA.cons, mA, s - 1)] - 1 : 15;
    int shB = (s >= 1 && s <= mB) ? (int)tabB[rc_at(SB.cons, mB, s - 1)] - 1 : 15;
    rsh[i] = mk(shA, shB);
    hg[i] = mk((s >= 1 && s < mA) ? -1 : 0, (s >= 1 && s < mB) ? -1 : 0);
    h[i] = 0;
    br[i] = 0;
    accA[i] = accB[i] = 0;
    hsnap[i] = 0;
  }
  const int T = nA + 63;
  const int nblk = (T + 15) >> 4;
  pk upPrev = 0, b = 0;
  int c = -lane;
  const pk two2 = 0x00020002u, m1 = 0xFFFFFFFFu;
  for (int blk = 0; blk < nblk; ++blk) {
    int ci = blk * 16 + (lane & 15);
    uint32_t chunk = 0;
    if (ci < nA) chunk = onehot(tabA, rc_at(SA.ref, nA, ci));
    if (ci < nB) chunk |= onehot(tabB, rc_at(SB.ref, nB, ci)) << 16;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      b = (pk)dpp_from_prev((int)b, __builtin_amdgcn_readlane((int)chunk, f));
      pk recv = dppz_from_prev(h[K - 1]);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)nA) {
        pk diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          pk sc = pk_shr(b, rsh[i]) & two2;
          pk x = pk_add(diag, sc);
          pk z = pk_add(h[i], hg[i]);
          pk nv = pk_max(pk_max(x, z), up);   // (x, z do not depend on the row above: one dependent op per row)
          diag = h[i];
          up = nv;
          h[i] = nv;
          pk d = pk_sub(nv, br[i]);
          br[i] = pk_max(br[i], nv);
          pk dm = pk_max(d, m1);
          if (f < 8) accA[i] = pk_add(accA[i], pk_shl_c(dm, 2 * f));
          else accB[i] = pk_add(accB[i], pk_shl_c(dm, 2 * (f - 8)));
        }
        if (c == nB) {  // B's real last column: rev[.][nB] of every slot
#pragma unroll
          for (int i = 0; i < K; ++i) hsnap[i] = h[i];
        }
      }
      upPrev = recv;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      stack[((size_t)(blk * 2 + 0) * K + i) * WAVE + lane] = pk_add(accA[i], 0x55555555u);
      stack[((size_t)(blk * 2 + 1) * K + i) * WAVE + lane] = pk_add(accB[i], 0x55555555u);
      accA[i] = accB[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    hfin[i] = h[i];
    brfin[i] = br[i];
  }
}

// packed M-pass with join.  Mirrored slots (slot s = row m - s), lane 63 leads; B starts
// delta = nA - nB columns late.  Outputs per slot: best sum' (16 bit) and the column of its
// last strict improvement, and the final V' of slot 0 (= row m) of both junctions.
template <int K>
__device__ __forceinline__ void pass_M2(const StrLdsFwd& SA, const StrLdsFwd& SB, const uint8_t* tabA, const uint8_t* tabB,
                                        int mA, int nA, int mB, int nB, const uint32_t* stack, int lane,
                                        const pk (&brfin)[K], pk (&best)[K], pk (&bestc)[K], pk& hrow_m) {
  pk rsh[K], hg[K], h[K], bm[K], g[K], dwA[K], dwB[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    int rA = mA - s, rB = mB - s;
    int shA = (rA >= 1) ? (int)tabA[SA.cons[rA - 1]] - 1 : 15;
    int shB = (rB >= 1) ? (int)tabB[SB.cons[rB - 1]] - 1 : 15;
    rsh[i] = mk(shA, shB);
    hg[i] = mk((rA >= 1 && rA < mA) ? -1 : 0, (rB >= 1 && rB < mB) ? -1 : 0);
    h[i] = 0;
    bm[i] = 0;
    g[i] = mk((rA >= 0) ? lo16(brfin[i]) : NEG16, (rB >= 0) ? hi16(brfin[i]) : NEG16);
    best[i] = NEG2;
    bestc[i] = 0;
    dwA[i] = dwB[i] = 0;
  }
  const int T = nA + 63;
  const int nblk = (T + 15) >> 4;
  const int delta = nA - nB;
  pk upPrev = 0, b = 0;
  int c = (T - nblk * 16) - 63 + lane;
  const pk two2 = 0x00020002u, three2 = 0x00030003u;
  for (int blk = nblk - 1; blk >= 0; --blk) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      uint32_t w0 = ld_scratch(&stack[((size_t)(blk * 2 + 0) * K + i) * WAVE + lane]);
      uint32_t w1 = ld_scratch(&stack[((size_t)(blk * 2 + 1) * K + i) * WAVE + lane]);
      // codes {0 below,1 tie,2 +1,3 +2} -> delta fields {0,0,1,2}
      uint32_t hi0 = (w0 >> 1) & 0x55555555u, lo0 = w0 & 0x55555555u;
      dwA[i] = (hi0 & ~lo0) | ((hi0 & lo0) << 1);
      uint32_t hi1 = (w1 >> 1) & 0x55555555u, lo1 = w1 & 0x55555555u;
      dwB[i] = (hi1 & ~lo1) | ((hi1 & lo1) << 1);
    }
    int ci = T - blk * 16 - 16 + (lane & 15);  // A-numbered column index - 1 of step f = 15 - (lane&15)
    uint32_t chunk = 0;
    if (ci >= 0 && ci < nA) chunk = onehot(tabA, SA.ref[ci]);
    {
      int cb = ci - delta;
      if (cb >= 0 && cb < nB) chunk |= onehot(tabB, SB.ref[cb]) << 16;
    }
#pragma unroll
    for (int f = 15; f >= 0; --f) {
      b = (pk)dpp_from_next((int)b, __builtin_amdgcn_readlane((int)chunk, 15 - f));
      pk recv = dppz_from_next(h[0]);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)nA) {
        const int cB = c - delta;
        if (c == 1 || cB == 1) {  // first real column of A / B in this lane: column-0 candidate
          const pk mask = ((c == 1) ? 0x0000FFFFu : 0u) | ((cB == 1) ? 0xFFFF0000u : 0u);
#pragma unroll
          for (int i = 0; i < K; ++i) {
            pk cand = pk_add(bm[i], g[i]);
            best[i] = (best[i] & ~mask) | (cand & mask);
            bestc[i] = bestc[i] & ~mask;
          }
        }
        const pk cpk = mk(c, cB);
        pk diag = upPrev, up = recv;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
          pk sc = pk_shr(b, rsh[i]) & two2;
          pk x = pk_add(diag, sc);
          pk z = pk_add(h[i], hg[i]);
          pk nv = pk_max(pk_max(x, z), up);   // (x, z do not depend on the row above: one dependent op per row)
          diag = h[i];
          up = nv;
          h[i] = nv;
          bm[i] = pk_max(bm[i], nv);
          pk dl = (((f < 8) ? dwA[i] : dwB[i]) >> (2 * (f & 7))) & three2;
          g[i] = pk_sub(g[i], dl);
          pk sum = pk_add(bm[i], g[i]);
          pk nb = pk_max(best[i], sum);
          pk mask = pk_sar15(pk_sub(best[i], nb));   // 0xFFFF where the sum strictly improved
          bestc[i] = (bestc[i] & ~mask) | (cpk & mask);
          best[i] = nb;
        }
      }
      upPrev = recv;
    }
  }
  hrow_m = h[0];
}

// ---- two junctions per wavefront: DP kernel ---------------------------------------------
// setup -> packed R-pass -> packed M-pass/join -> finish.  The join result (score_best,
// cons_left, ref_left, ref_right) is left in the result record, which split_post_kernel reads;
// junctions whose consensus has more than 15 distinct bytes get status = DH_DEFERRED and are
// handled by the 32-bit kernel (split_align_kernel) launched right after.
template <int K>
__device__ __forceinline__ void process_pair(const SplitArgs& A, int jA, int jB, PairLds& L, uint32_t* scratch,
                                             int lane) {
  JCtx X0, X1;
  junction_setup<K, true, StrLdsFwd>(A, jA, L.s[0], X0, lane);
  if (jB >= 0) junction_setup<K, true, StrLdsFwd>(A, jB, L.s[1], X1, lane);
  else {
    X1 = X0;
    X1.go = false;
    X1.m = 0;
    X1.n = 0;
  }
  jA = rfl(jA);
  jB = rfl(jB);
  if (!X0.go && !X1.go) return;
  // role A (lo half) = the junction with the longer reference among those that run
  const bool swap = !X0.go || (X1.go && X1.n > X0.n);
  JCtx XA = swap ? X1 : X0, XB = swap ? X0 : X1;
  StrLdsFwd& SA = swap ? L.s[1] : L.s[0];
  StrLdsFwd& SB = swap ? L.s[0] : L.s[1];
  uint8_t* tabA0 = swap ? L.tab[1][0] : L.tab[0][0];
  uint8_t* tabA1 = swap ? L.tab[1][1] : L.tab[0][1];
  uint8_t* tabB0 = swap ? L.tab[0][0] : L.tab[1][0];
  uint8_t* tabB1 = swap ? L.tab[0][1] : L.tab[1][1];
  const int mA = XA.m, nA = XA.n;
  const int mB = XB.go ? XB.m : 0, nB = XB.go ? XB.n : 0;
  int cntmax = 0;
  cntmax = max(cntmax, build_table<false>(SA.cons, mA, tabA0, lane));
  cntmax = max(cntmax, build_table<true>(SA.cons, mA, tabA1, lane));
  cntmax = max(cntmax, build_table<false>(SB.cons, mB, tabB0, lane));
  cntmax = max(cntmax, build_table<true>(SB.cons, mB, tabB1, lane));
  cntmax = rfl(cntmax);
  if (cntmax > 15) {
    // more than 15 distinct bytes in a consensus: cannot be one-hot coded in 16 bits
    if (lane == 0) {
      if (XA.go) XA.out->status = DH_DEFERRED;
      if (XB.go) XB.out->status = DH_DEFERRED;
      atomicAdd(A.work_counter, 1);  // tells split_align_kernel that it has work
    }
    return;
  }
  pk hfin[K], hsnap[K], brfin[K], best[K], bestc[K];
  pk hrow_m;
  pass_R2<K>(SA, SB, tabA1, tabB1, mA, nA, mB, nB, scratch, lane, hfin, hsnap, brfin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pass_M2<K>(SA, SB, tabA0, tabB0, mA, nA, mB, nB, scratch, lane, brfin, best, bestc, hrow_m);
  int rA = 0, rB = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (lane * K + i == mA) rA = lo16(hfin[i]);
    if (lane * K + i == mB) rB = hi16(hsnap[i]);
  }
  const int revA = __shfl(rA, mA / K) - mA, revB = __shfl(rB, mB / K) - mB;
  const pk h0 = (pk)__shfl((int)hrow_m, 0);
  const int unsA = lo16(h0) - mA, unsB = hi16(h0) - mB;
  long long kA = (long long)0x8000000000000000ll, kB = kA;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s = lane * K + i;
    if (s <= mA) {
      long long kk = ((long long)lo16(best[i]) << 32) | ((long long)s << 12) | (long long)(4095 - (int)(bestc[i] & 0xffffu));
      kA = kk > kA ? kk : kA;
    }
    if (s <= mB) {
      long long kk = ((long long)hi16(best[i]) << 32) | ((long long)s << 12) | (long long)(4095 - (int)(bestc[i] >> 16));
      kB = kk > kB ? kk : kB;
    }
  }
  kA = wave_max64(kA);
  kB = wave_max64(kB);
  // finish both roles in a real (not unrolled) loop: two inlined copies of junction_finish were
  // tail-merged by hipcc into an exec-mask "loop" that never terminated on gfx950
  JCtx XX[2] = {XA, XB};
  const int uns2[2] = {unsA, unsB}, rev2[2] = {revA, revB};
  const long long key2[2] = {kA, kB};
#pragma unroll 1
  for (int role = 0; role < 2; ++role) {
    if (!XX[role].go) continue;
    const int sh = 16 * role;
    auto code_word = [&](int slot, int t) -> uint32_t {
      int ls = slot / K, is = slot - ls * K;
      return (ld_scratch(&scratch[((size_t)(t >> 3) * K + is) * WAVE + ls]) >> sh) & 0xffffu;
    };
    junction_finish<K>(XX[role], uns2[role], rev2[role], key2[role], code_word, 8, lane);
  }
}

#ifndef DH_PAIR_WAVES
#define DH_PAIR_WAVES 5
#endif
template <int K>
__global__ __launch_bounds__(WAVE, DH_PAIR_WAVES) void split_pair_kernel(SplitArgs A0) {
  __shared__ PairLds L;
  const int lane = threadIdx.x;
  if (A0.sps_left && *A0.sps_left == 0) return;   // the sparse kernel finished every junction of the batch
  const SplitArgs A = A0;   // (the copy the called helpers read: made behind the early exit, see split_align_kernel)
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
      if (w >= A.n_work) break;
    int ja = A.work_list[2 * w], jb = A.work_list[2 * w + 1];   // (junctions finished by the sparse kernel: empty seats)
    if (ja >= 0 && rfl(A.res[ja].reserved) == SPS_DONE) ja = -1;
    if (jb >= 0 && rfl(A.res[jb].reserved) == SPS_DONE) jb = -1;
    if (ja < 0) { ja = jb; jb = -1; }
    if (ja >= 0) process_pair<K>(A, ja, jb, L, scratch, lane);
    }
}

}  // namespace dh
