// split_quad.hpp -- packed longNeedle passes: FOUR junctions per 64-lane wavefront.
//
// The packed pair kernel (split_pk.hpp) gives every junction pair all 64 lanes, K rows per lane.
// A 150 bp consensus needs 151 slots: K = 3 leaves 41 of 192 slots (21 %) idle and the 63-step
// lane skew adds 6 % to a 1000-column sweep.  Here each HALF of the wavefront (32 lanes, K <= 5
// rows per lane, |consensus| <= 32*K - 1 <= 159) carries its own packed pair: lanes 0-31 hold
// junctions (A lo, B hi), lanes 32-63 hold (C lo, D hi).  151 of 160 slots are busy, the skew is
// 31 steps, and the per-step overhead (letters, hand-off, loop control) is shared by four
// junctions.  Same arithmetic, code words and result records as split_pk.hpp; the half-wave seam
// costs two v_writelane per step (the DPP wave shift crosses lane 31 -> 32 and must be cut there).
#pragma once
#include "split_pk.hpp"

namespace dh {

constexpr int HALF = 32;
constexpr int QNMAX = 1280;   // |svRefStr| limit of the quad kernel (LDS budget: 4 junctions per wavefront)

struct __attribute__((aligned(16))) StrLdsQ {
  static constexpr bool has_rc = false;
  static constexpr int ref_cap = QNMAX;   // (the host only seats junctions whose window bound is <= QNMAX)
  static constexpr int cons_cap = HALF * 5;
  uint8_t cons[HALF * 5];
  uint8_t ref[QNMAX];
};

struct __attribute__((aligned(16))) QuadLds {
  StrLdsQ s[4];
  uint8_t tab[4][2][256];
};

// v_writelane_b32 of a wave-uniform value into one lane (compile-time lane index)
template <int LANE>
__device__ __forceinline__ pk wl_const(pk v, int sval) {
  int r = (int)v;
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(r) : "s"(sval), "n"(LANE));
  return (pk)r;
}

// packed R-pass, two pairs.  Per-lane quantities (mLo, nLo, mHi, nHi, string / table pointers)
// belong to the lane's half.  Same stack layout as pass_R2 (indexed by the absolute lane).
template <int K>
__device__ __forceinline__ void pass_R4(const StrLdsQ* SLo, const StrLdsQ* SHi, const uint8_t* tabLo, const uint8_t* tabHi,
                                        int mLo, int nLo, int mHi, int nHi, int nMax, uint32_t* stack, int lane,
                                        pk (&hfin)[K], pk (&hsnap)[K], pk (&brfin)[K]) {
  const int l5 = lane & (HALF - 1);
  pk rsh[K], hg[K], h[K], br[K], accA[K], accB[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int s = l5 * K + i;
    const int shA = (s >= 1 && s <= mLo) ? (int)tabLo[rc_at(SLo->cons, mLo, s - 1)] - 1 : 15;
    const int shB = (s >= 1 && s <= mHi) ? (int)tabHi[rc_at(SHi->cons, mHi, s - 1)] - 1 : 15;
    rsh[i] = mk(shA, shB);
    hg[i] = mk((s >= 1 && s < mLo) ? -1 : 0, (s >= 1 && s < mHi) ? -1 : 0);
    h[i] = 0;
    br[i] = 0;
    accA[i] = accB[i] = 0;
    hsnap[i] = 0;
  }
  const int T = nMax + HALF - 1;
  const int nblk = (T + 15) >> 4;
  pk upPrev = 0, b = 0;
  int c = -l5;
  const pk two2 = 0x00020002u, m1 = 0xFFFFFFFFu;
  for (int blk = 0; blk < nblk; ++blk) {
    const int ci = blk * 16 + (lane & 15);
    uint32_t chunk = 0;
    if (ci < nLo) chunk = onehot(tabLo, rc_at(SLo->ref, nLo, ci));
    if (ci < nHi) chunk |= onehot(tabHi, rc_at(SHi->ref, nHi, ci)) << 16;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      b = (pk)dpp_from_prev((int)b, __builtin_amdgcn_readlane((int)chunk, f));
      b = wl_const<HALF>(b, __builtin_amdgcn_readlane((int)chunk, HALF + f));   // first lane of the upper half
      pk recv = dppz_from_prev(h[K - 1]);
      recv = wl_const<HALF>(recv, 0);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)nLo) {
        pk diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const pk sc = pk_shr(b, rsh[i]) & two2;
          const pk x = pk_add(diag, sc);
          const pk z = pk_add(h[i], hg[i]);
          const pk nv = pk_max(pk_max(x, z), up);   // (x, z do not depend on the row above: one dependent op per row)
          diag = h[i];
          up = nv;
          h[i] = nv;
          const pk d = pk_sub(nv, br[i]);
          br[i] = pk_max(br[i], nv);
          const pk dm = pk_max(d, m1);
          if (f < 8) accA[i] = pk_add(accA[i], pk_shl_c(dm, 2 * f));
          else accB[i] = pk_add(accB[i], pk_shl_c(dm, 2 * (f - 8)));
        }
        if (c == nHi) {
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

// packed M-pass with join, two pairs; mirrored slots inside each half (lanes 31 and 63 lead)
template <int K>
__device__ __forceinline__ void pass_M4(const StrLdsQ* SLo, const StrLdsQ* SHi, const uint8_t* tabLo, const uint8_t* tabHi,
                                        int mLo, int nLo, int mHi, int nHi, int nMax, const uint32_t* stack, int lane,
                                        const pk (&brfin)[K], pk (&best)[K], pk (&bestc)[K], pk& hrow_m) {
  const int l5 = lane & (HALF - 1);
  pk rsh[K], hg[K], h[K], bm[K], g[K], dwA[K], dwB[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int s = l5 * K + i;
    const int rA = mLo - s, rB = mHi - s;
    const int shA = (rA >= 1) ? (int)tabLo[SLo->cons[rA - 1]] - 1 : 15;
    const int shB = (rB >= 1) ? (int)tabHi[SHi->cons[rB - 1]] - 1 : 15;
    rsh[i] = mk(shA, shB);
    hg[i] = mk((rA >= 1 && rA < mLo) ? -1 : 0, (rB >= 1 && rB < mHi) ? -1 : 0);
    h[i] = 0;
    bm[i] = 0;
    g[i] = mk((rA >= 0) ? lo16(brfin[i]) : NEG16, (rB >= 0) ? hi16(brfin[i]) : NEG16);
    best[i] = NEG2;
    bestc[i] = 0;
    dwA[i] = dwB[i] = 0;
  }
  const int T = nMax + HALF - 1;
  const int nblk = (T + 15) >> 4;
  const int delta = nLo - nHi;
  pk upPrev = 0, b = 0;
  int c = nLo - 16 * nblk + l5;   // column of this lane one step before the first consumer step
  const pk two2 = 0x00020002u, three2 = 0x00030003u;
  for (int blk = nblk - 1; blk >= 0; --blk) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const uint32_t w0 = ld_scratch(&stack[((size_t)(blk * 2 + 0) * K + i) * WAVE + lane]);
      const uint32_t w1 = ld_scratch(&stack[((size_t)(blk * 2 + 1) * K + i) * WAVE + lane]);
      const uint32_t hi0 = (w0 >> 1) & 0x55555555u, lo0 = w0 & 0x55555555u;
      dwA[i] = (hi0 & ~lo0) | ((hi0 & lo0) << 1);
      const uint32_t hi1 = (w1 >> 1) & 0x55555555u, lo1 = w1 & 0x55555555u;
      dwB[i] = (hi1 & ~lo1) | ((hi1 & lo1) << 1);
    }
    // leader's (l5 = 31) column at step f is nLo - (16 blk + f) + 31; lane j = 15 - f of each half's first
    // 16 lanes fetches its letter
    const int ci = nLo + 15 - blk * 16 + (lane & 15);
    uint32_t chunk = 0;
    if (ci >= 0 && ci < nLo) chunk = onehot(tabLo, SLo->ref[ci]);
    {
      const int cb = ci - delta;
      if (cb >= 0 && cb < nHi) chunk |= onehot(tabHi, SHi->ref[cb]) << 16;
    }
#pragma unroll
    for (int f = 15; f >= 0; --f) {
      b = (pk)dpp_from_next((int)b, __builtin_amdgcn_readlane((int)chunk, HALF + 15 - f));   // lane 63 <- upper half's letter
      b = wl_const<HALF - 1>(b, __builtin_amdgcn_readlane((int)chunk, 15 - f));               // lane 31 <- lower half's letter
      pk recv = dppz_from_next(h[0]);
      recv = wl_const<HALF - 1>(recv, 0);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)nLo) {
        const int cB = c - delta;
        if (c == 1 || cB == 1) {
          const pk mask = ((c == 1) ? 0x0000FFFFu : 0u) | ((cB == 1) ? 0xFFFF0000u : 0u);
#pragma unroll
          for (int i = 0; i < K; ++i) {
            const pk cand = pk_add(bm[i], g[i]);
            best[i] = (best[i] & ~mask) | (cand & mask);
            bestc[i] = bestc[i] & ~mask;
          }
        }
        const pk cpk = mk(c, cB);
        pk diag = upPrev, up = recv;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
          const pk sc = pk_shr(b, rsh[i]) & two2;
          const pk x = pk_add(diag, sc);
          const pk z = pk_add(h[i], hg[i]);
          const pk nv = pk_max(pk_max(x, z), up);   // (x, z do not depend on the row above: one dependent op per row)
          diag = h[i];
          up = nv;
          h[i] = nv;
          bm[i] = pk_max(bm[i], nv);
          const pk dl = (((f < 8) ? dwA[i] : dwB[i]) >> (2 * (f & 7))) & three2;
          g[i] = pk_sub(g[i], dl);
          const pk sum = pk_add(bm[i], g[i]);
          const pk nb = pk_max(best[i], sum);
          const pk mask = pk_sar15(pk_sub(best[i], nb));
          bestc[i] = (bestc[i] & ~mask) | (cpk & mask);
          best[i] = nb;
        }
      }
      upPrev = recv;
    }
  }
  hrow_m = h[0];
}

// max of a signed 64-bit key over the 32 lanes of the caller's half
__device__ __forceinline__ long long half_max64(long long v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) {
    const int lo = __shfl_xor((int)(v & 0xffffffffll), o);
    const int hi = __shfl_xor((int)(v >> 32), o);
    const long long w = ((long long)hi << 32) | (unsigned int)lo;
    v = (w > v) ? w : v;
  }
  return v;
}

template <int K>
__device__ __forceinline__ void process_quad(const SplitArgs& A, const int (&jq)[4], QuadLds& L, uint32_t* scratch, int lane) {
  JCtx X[4];
  bool any = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (jq[q] >= 0) {
      junction_setup<K, true, StrLdsQ>(A, jq[q], L.s[q], X[q], lane);
      if (X[q].go && (X[q].m + 1 > HALF * K || X[q].n > QNMAX)) {   // does not fit a half wavefront: 32-bit kernel
        if (lane == 0) {
          X[q].out->status = DH_DEFERRED;
          atomicAdd(A.work_counter + 16, 1);
        }
        X[q].go = false;
      }
    } else {
      X[q] = X[0];
      X[q].go = false;
      X[q].m = 0;
      X[q].n = 0;
    }
    any = any || X[q].go;
  }
  if (!any) return;
  // per half: role lo = the junction with the longer reference among those that run
  int idxLo[2], idxHi[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int a = 2 * hf, bq = 2 * hf + 1;
    const bool swap = !X[a].go || (X[bq].go && X[bq].n > X[a].n);
    idxLo[hf] = swap ? bq : a;
    idxHi[hf] = swap ? a : bq;
  }
  int cntmax = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mq = X[q].go ? X[q].m : 0;
    cntmax = max(cntmax, build_table<false>(L.s[q].cons, mq, L.tab[q][0], lane));
    cntmax = max(cntmax, build_table<true>(L.s[q].cons, mq, L.tab[q][1], lane));
  }
  cntmax = rfl(cntmax);
  if (cntmax > 15) {   // a consensus with more than 15 distinct bytes cannot be one-hot coded in 16 bits
    if (lane == 0) {
      for (int q = 0; q < 4; ++q)
        if (X[q].go) X[q].out->status = DH_DEFERRED;
      atomicAdd(A.work_counter + 16, 1);
    }
    return;
  }
  const int hf = lane >> 5;
  const int qLo = hf ? idxLo[1] : idxLo[0], qHi = hf ? idxHi[1] : idxHi[0];
  const StrLdsQ* SLo = &L.s[qLo];
  const StrLdsQ* SHi = &L.s[qHi];
  const uint8_t* tabLo0 = L.tab[qLo][0];
  const uint8_t* tabLo1 = L.tab[qLo][1];
  const uint8_t* tabHi0 = L.tab[qHi][0];
  const uint8_t* tabHi1 = L.tab[qHi][1];
  const int m0 = X[idxLo[0]].go ? X[idxLo[0]].m : 0, n0 = X[idxLo[0]].go ? X[idxLo[0]].n : 0;
  const int m1 = X[idxHi[0]].go ? X[idxHi[0]].m : 0, n1 = X[idxHi[0]].go ? X[idxHi[0]].n : 0;
  const int m2 = X[idxLo[1]].go ? X[idxLo[1]].m : 0, n2 = X[idxLo[1]].go ? X[idxLo[1]].n : 0;
  const int m3 = X[idxHi[1]].go ? X[idxHi[1]].m : 0, n3 = X[idxHi[1]].go ? X[idxHi[1]].n : 0;
  const int mLo = hf ? m2 : m0, nLo = hf ? n2 : n0, mHi = hf ? m3 : m1, nHi = hf ? n3 : n1;
  const int nMax = max(n0, n2);
  pk hfin[K], hsnap[K], brfin[K], best[K], bestc[K];
  pk hrow_m;
  pass_R4<K>(SLo, SHi, tabLo1, tabHi1, mLo, nLo, mHi, nHi, nMax, scratch, lane, hfin, hsnap, brfin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pass_M4<K>(SLo, SHi, tabLo0, tabHi0, mLo, nLo, mHi, nHi, nMax, scratch, lane, brfin, best, bestc, hrow_m);
  const int l5 = lane & (HALF - 1), base = lane & HALF;
  int rA = 0, rB = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (l5 * K + i == mLo) rA = lo16(hfin[i]);
    if (l5 * K + i == mHi) rB = hi16(hsnap[i]);
  }
  const int revLo = __shfl(rA, base + mLo / K) - mLo, revHi = __shfl(rB, base + mHi / K) - mHi;
  const pk h0 = (pk)__shfl((int)hrow_m, base);
  const int unsLo = lo16(h0) - mLo, unsHi = hi16(h0) - mHi;
  long long kLo = (long long)0x8000000000000000ll, kHi = kLo;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int s = l5 * K + i;
    if (s <= mLo) {
      const long long kk = ((long long)lo16(best[i]) << 32) | ((long long)s << 12) | (long long)(4095 - (int)(bestc[i] & 0xffffu));
      kLo = kk > kLo ? kk : kLo;
    }
    if (s <= mHi) {
      const long long kk = ((long long)hi16(best[i]) << 32) | ((long long)s << 12) | (long long)(4095 - (int)(bestc[i] >> 16));
      kHi = kk > kHi ? kk : kHi;
    }
  }
  kLo = half_max64(kLo);
  kHi = half_max64(kHi);
  // four roles, finished one after the other in a real loop (see process_pair)
  JCtx XX[4];
  int uns4[4], rev4[4];
  long long key4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int src = (r >> 1) * HALF;   // any lane of the role's half holds its values
    const int q = (r & 1) ? ((r >> 1) ? idxHi[1] : idxHi[0]) : ((r >> 1) ? idxLo[1] : idxLo[0]);
    XX[r] = X[q];
    uns4[r] = __builtin_amdgcn_readlane((r & 1) ? unsHi : unsLo, src);
    rev4[r] = __builtin_amdgcn_readlane((r & 1) ? revHi : revLo, src);
    const long long kk = (r & 1) ? kHi : kLo;
    key4[r] = ((long long)__builtin_amdgcn_readlane((int)(kk >> 32), src) << 32) |
              (unsigned int)__builtin_amdgcn_readlane((int)(kk & 0xffffffffll), src);
  }
#pragma unroll 1
  for (int role = 0; role < 4; ++role) {
    if (!XX[role].go) continue;
    const int sh = 16 * (role & 1), lbase = (role >> 1) * HALF;
    auto code_word = [&](int slot, int t) -> uint32_t {
      const int ls = slot / K, is = slot - ls * K;
      return (ld_scratch(&scratch[((size_t)(t >> 3) * K + is) * WAVE + lbase + ls]) >> sh) & 0xffffu;
    };
    junction_finish<K>(XX[role], uns4[role], rev4[role], key4[role], code_word, 8, lane);
  }
}

#ifndef DH_QUAD_WAVES
#define DH_QUAD_WAVES 4
#endif
// One launch, two item kinds: the first n_quads work items seat four junctions per wavefront (4 indices
// each), the remaining items are packed pairs (2 indices each, process_pair<KP>).  A batch of N junctions
// gives only N/4 quad wavefronts of ~1 ms; when that is not a multiple of the number of SIMDs the last
// round runs the chip partly empty, so the host tops a whole number of quad rounds up with pair items
// (cheaper wavefronts) instead of a thin extra quad round.  -1 = empty seat.
template <int KQ, int KP>
__global__ __launch_bounds__(WAVE, DH_QUAD_WAVES) void split_quad_kernel(SplitArgs A0, int n_quads) {
  __shared__ union {
    QuadLds q;
    PairLds p;
  } L;
  const int lane = threadIdx.x;
  if (A0.sps_left && *A0.sps_left == 0) return;   // the sparse kernel finished every junction of the batch
  const SplitArgs A = A0;   // (the copy the called helpers read: made behind the early exit, see split_align_kernel)
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    if (w < n_quads) {
      // seats whose junction the sparse kernel already finished are empty; live junctions move to the front
      int jq[4] = {-1, -1, -1, -1};
      int nlive = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = A.work_list[4 * w + q];
        if (j >= 0 && rfl(A.res[j].reserved) != SPS_DONE) {
          if (nlive == 0) jq[0] = j;
          else if (nlive == 1) jq[1] = j;
          else if (nlive == 2) jq[2] = j;
          else jq[3] = j;
          ++nlive;
        }
      }
      if (nlive > 0) process_quad<KQ>(A, jq, L.q, scratch, lane);
    } else {
      const int32_t* pl = A.work_list + 4 * n_quads + 2 * (w - n_quads);
      int ja = pl[0], jb = pl[1];
      if (ja >= 0 && rfl(A.res[ja].reserved) == SPS_DONE) ja = -1;
      if (jb >= 0 && rfl(A.res[jb].reserved) == SPS_DONE) jb = -1;
      if (ja < 0) { ja = jb; jb = -1; }
      if (ja >= 0) process_pair<KP>(A, ja, jb, L.p, scratch, lane);
    }
    __syncthreads();
  }
}

}  // namespace dh
