// lrins_kernel.hpp -- gfx950 device code for alignConsensus() of long-read INSERTION junctions
// (svt 4 with |consensus| > 319 or |svRefStr| > 2048): splitAlign (src/split.h:480-538) whose six
// edlibAlign calls run in edlib's Hirschberg regime.  Same flow as ins_kernel.hpp, on the strip
// machinery of lrmsa_kernel.hpp (lm_hw / lm_shw / lm_nw_path reproduce edlib's HW / SHW / NW
// results including the Hirschberg split), strings and op strings in the wavefront's HBM
// workspace, column masks in LDS, shared split_detect stage.
#pragma once
#include "lrmsa_kernel.hpp"

namespace dh {

struct LrInsArgs {
  uint8_t* ws;
  uint64_t ws_stride;
  int32_t mcap, ncap;
  uint64_t off_rcons, off_ref, off_rref, off_bnd, off_opsL, off_opsR, off_tmp, off_dist, off_dirs;   // cons at 0
  uint64_t strip_words;
  int32_t realign;        // src/split.h:564-572
};

// editDistanceVec (split.h:377-405) on a FORWARD op string
__device__ __forceinline__ void lri_edit_distance_vec(const uint8_t* ops, int nops, int32_t* dist, int lane) {
  int qbase = 0, ebase = 0;
  const unsigned long long le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  for (int k = 0; k < nops; k += WAVE) {
    const int idx = k + lane;
    const int op = (idx < nops) ? (int)ops[idx] : ED_DELETE;
    const bool isq = (idx < nops) && (op != ED_DELETE);
    const bool ise = (idx < nops) && (op != ED_MATCH);
    const unsigned long long bq = __ballot(isq), be = __ballot(ise);
    if (isq) dist[qbase + __popcll(bq & le) - 1] = ebase + __popcll(be & le);
    qbase += __popcll(bq);
    ebase += __popcll(be);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <typename PL>
__device__ __forceinline__ int lri_mask_append_ops(PL& L, int pos, const uint8_t* ops, int nops, int lane) {
  for (int k = 0; k < nops; k += WAVE) {
    const int idx = k + lane;
    const int op = (idx < nops) ? (int)ops[idx] : 0;
    const unsigned long long v = __ballot(idx < nops && op != ED_INSERT);
    const unsigned long long r = __ballot(idx < nops && op != ED_DELETE);
    const int cnt = min(WAVE, nops - k);
    mask_append(L, pos, cnt, v, r, lane);
    pos += cnt;
  }
  return pos;
}
template <typename PL>
__device__ __forceinline__ int lri_mask_append_run(PL& L, int pos, int cnt, unsigned long long v, unsigned long long r, int lane) {
  for (int k = 0; k < cnt; k += WAVE) {
    mask_append(L, pos, min(WAVE, cnt - k), v, r, lane);
    pos += min(WAVE, cnt - k);
  }
  return pos;
}

__device__ void process_lr_ins(const SplitArgs& A, const LrInsArgs& R, int j, PostLR& L, MyersLds<MYERS_NW>& ML, uint8_t* ws, int lane) {
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
  X.direct = (A.ref_base != nullptr);   // dellyhip_split_align beyond the short-read shapes: svRefStr given, splitAlign only (round 6)
  X.consLeft = X.refLeft = X.refRight = X.consRight = 0;
  StrPtr S{ws, ws + R.off_rcons, ws + R.off_ref, ws + R.off_rref};
  int32_t* bnd = reinterpret_cast<int32_t*>(ws + R.off_bnd);
  const int bnd_stride = R.ncap + 128;
  uint8_t* opsL = ws + R.off_opsL;
  uint8_t* opsR = ws + R.off_opsR;
  uint8_t* tmp = ws + R.off_tmp;
  int32_t* distF = reinterpret_cast<int32_t*>(ws + R.off_dist);
  int32_t* distR = distF + (R.ncap + 64);
  uint32_t* dirs = reinterpret_cast<uint32_t*>(ws + R.off_dirs);
  const int ops_cap = R.mcap + R.ncap + 32;
  const int m = X.m;
  const uint8_t* cons_g = A.cons_base + A.cons_off[j];
  const bool own_cons = (A.cons_base != A.out_blob) || (cons_g == X.ob);
  const int prior = X.out->status, support = X.out->sr_support;
  int status = 0;
  bool go = true, mlimit = false;
  if (prior) { status = prior; mlimit = true; go = false; }
  else if (m < 1 || m > LR_MMAX || m > R.mcap) { status = DELLYHIP_E_LIMIT; mlimit = (m != 0); go = false; }
  if (go) {
    for (int i = lane; i < m; i += WAVE) {
      const uint8_t ch = cons_g[i];
      S.cons[i] = ch;
      if (A.cons_base != A.out_blob) X.ob[i] = ch;   // (MSA modes: the consensus already lives in the slot)
    }
  }
  if (go && !X.direct && !(P.reserved & 2) && m < 2 * P.minimum_flank_size + J.ins_len) go = false;     // split.h:647
  Seg seg[3];
  int nseg = 0, n = 0;
  if (go && X.direct) {
    n = A.ref_len[j];
    if (n > LR_NMAX || n > R.ncap || n < 3) { status = DELLYHIP_E_LIMIT; go = false; n = 0; }
    else {
      const uint8_t* rg = A.ref_base + A.ref_off[j];
      for (int i = lane; i < n; i += WAVE) S.ref[i] = rg[i];
    }
  } else if (go) {
    int sBeg, sEnd, eBeg, eEnd;
    (void)window_segments<true>(A, J, m, seg, nseg, sBeg, sEnd, eBeg, eEnd);
    X.sBeg = sBeg; X.sEnd = sEnd; X.eBeg = eBeg; X.eEnd = eEnd;
    n = seg[0].len;
    if (n > LR_NMAX || n > R.ncap || n < 3) { status = DELLYHIP_E_LIMIT; go = false; }
    if (go) fill_segment(S.ref, seg[0], lane);
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
  if (go && R.realign) {
    // split.h:564-572: keep the orientation with the smaller NW edit distance to the window
    for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int dF, dR;   // (pattern = the shorter string; strips beyond MYERS_ROWS rows, see lr_kernel.hpp)
    if (min(m, n) <= MYERS_ROWS) {
      if (m <= n) {
        dF = rfl(myers_nw_auto(ML, S.cons, m, S.ref, n, lane));   // (pattern bytes are fetched 32 at a time: the strings sit inside the workspace)
        dR = rfl(myers_nw_auto(ML, S.rcons, m, S.ref, n, lane));
      } else {
        dF = rfl(myers_nw_auto(ML, S.ref, n, S.cons, m, lane));
        dR = rfl(myers_nw_auto(ML, S.ref, n, S.rcons, m, lane));
      }
    } else {
      int8_t* hb0 = reinterpret_cast<int8_t*>(bnd);
      int8_t* hb1 = reinterpret_cast<int8_t*>(bnd + bnd_stride);
      if (m <= n) {
        dF = rfl(myers_nw_big(S.cons, m, S.ref, n, hb0, hb1, lane));
        dR = rfl(myers_nw_big(S.rcons, m, S.ref, n, hb0, hb1, lane));
      } else {
        dF = rfl(myers_nw_big(S.ref, n, S.cons, m, hb0, hb1, lane));
        dR = rfl(myers_nw_big(S.ref, n, S.rcons, m, hb0, hb1, lane));
      }
    }
    if (dR < dF) {
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
    for (int i = lane; i < n; i += WAVE) S.rref[i] = rc_at(S.ref, n, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  X.go = go;
  X.uniformize();
  int Ltot = 0;
  if (go) {
    const uint8_t* cons = S.cons;
    const uint8_t* ref = S.ref;
    // consensus and window made of A, C, G, T only (the usual case): the compare-free bit-vector passes give the same
    // distances / locations / op strings as plain byte equality (lm_pure_acgt)
    const int PM = (lm_pure_acgt(cons, 1, m, lane) && lm_pure_acgt(ref, 1, n, lane)) ? (LM_EQ | LM_EQFAST) : 0;
    // --- splitAlign, split.h:483-492
    const LmRes pre = lm_hw(cons, m, ref, n / 3, PM, true, false, bnd, bnd_stride, dirs, R.strip_words, tmp, opsL, ops_cap, lane);
    const uint32_t csStart = (uint32_t)pre.startLoc;
    const int so = (int)((2ull * (uint64_t)n) / 3ull);
    const LmRes suf = lm_hw(cons, m, ref + so, n - so, PM, false, false, bnd, bnd_stride, dirs, R.strip_words, tmp, opsL, ops_cap, lane);
    const uint32_t csEnd = (uint32_t)suf.endLoc;
    if (lane == 0) {
      X.out->score_unsplit = (int32_t)csStart;
      X.out->score_best = (int32_t)csEnd;
    }
    if (csStart >= csEnd) go = false;
    int bestJoin = 0;
    if (go) {
      uint32_t cslu = csEnd - csStart;
      if (cslu > (uint32_t)m - csStart) cslu = (uint32_t)m - csStart;
      const int csl = (int)cslu;
      const uint8_t* cs = cons + csStart;
      if (csl == 0) {
        for (int i = lane; i < n; i += WAVE) { distF[i] = 0; distR[i] = 0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        const LmRes f = lm_shw(cs, csl, ref, n, PM, bnd, bnd_stride, dirs, R.strip_words, tmp, opsL, ops_cap, lane);
        if (f.nops < 0) go = false;
        else lri_edit_distance_vec(opsL, f.nops, distF, lane);
        for (int i = lane; i < csl; i += WAVE) S.rcons[i] = rc_at(cs, csl, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (go) {
          const LmRes r = lm_shw(S.rcons, csl, S.rref, n, PM, bnd, bnd_stride, dirs, R.strip_words, tmp, opsL, ops_cap, lane);
          if (r.nops < 0) go = false;
          else lri_edit_distance_vec(opsL, r.nops, distR, lane);
        }
        if (!go && lane == 0) X.out->status = DELLYHIP_E_LIMIT;
      }
      if (go) {
        // best join, split.h:513-517: first minimum of distFwd[i] + distRev[n-i-2], i = 0..n-2
        unsigned long long key = ~0ull;
        for (int i = lane; i <= n - 2; i += WAVE) {
          const unsigned long long sum = (unsigned long long)(uint32_t)distF[i] + (unsigned long long)(uint32_t)distR[n - i - 2];
          const unsigned long long kk = (sum << 20) | (unsigned long long)i;
          key = kk < key ? kk : key;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          const int lo = __shfl_xor((int)(key & 0xffffffffull), o), hi = __shfl_xor((int)(key >> 32), o);
          const unsigned long long w = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
          key = w < key ? w : key;
        }
        bestJoin = rfl((int)(key & 0xfffffull));
        if (lane == 0) X.out->cons_left = bestJoin;
      }
    }
    LmRes le, ri;
    le.nops = ri.nops = 0;
    uint32_t gaplen = 0, missingStart = 0, missingEnd = 0;
    if (go) {
      le = lm_hw(cons, m, ref, bestJoin + 1, PM, true, true, bnd, bnd_stride, dirs, R.strip_words, tmp, opsL, ops_cap, lane);
      ri = lm_hw(cons, m, ref + bestJoin + 1, n - bestJoin - 1, PM, true, true, bnd, bnd_stride, dirs, R.strip_words, tmp, opsR,
                 ops_cap, lane);
      if (le.nops < 0 || ri.nops < 0) {
        if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
        go = false;
      }
    }
    if (go) {
      const uint32_t leftEnd = (uint32_t)le.endLoc, rightStart = (uint32_t)ri.startLoc;
      if (lane == 0) {
        X.out->ref_left = (int32_t)leftEnd;
        X.out->ref_right = (int32_t)rightStart;
      }
      if (leftEnd + 15u >= rightStart) go = false;  // split.h:532
      gaplen = rightStart - leftEnd - 1u;
      missingStart = (uint32_t)le.startLoc;
      missingEnd = (uint32_t)ri.endLoc;
      if (missingEnd < (uint32_t)m) missingEnd = (uint32_t)m - missingEnd - 1u;
      if (go) {
        const uint64_t total = (uint64_t)missingStart + (uint64_t)le.nops + gaplen + (uint64_t)ri.nops + missingEnd;
        if (total > (uint64_t)LR_MASKW * 64) {
          if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
          go = false;
        }
      }
    }
    go = rfl((int)go) != 0;
    if (go) {
      for (int w = lane; w < LR_MASKW; w += WAVE) {
        L.mV[w] = 0;
        L.mR[w] = 0;
        L.mE[w] = 0;
      }
      __syncthreads();
      int pos = 0;
      pos = lri_mask_append_run(L, pos, (int)missingStart, ~0ull, 0ull, lane);
      pos = lri_mask_append_ops(L, pos, opsL, le.nops, lane);
      pos = lri_mask_append_run(L, pos, (int)gaplen, ~0ull, 0ull, lane);
      pos = lri_mask_append_ops(L, pos, opsR, ri.nops, lane);
      pos = lri_mask_append_run(L, pos, (int)missingEnd, ~0ull, 0ull, lane);
      Ltot = pos;
      masks_finish(A, X, S, L, Ltot, Ltot, lane);
    }
  }
  X.go = go;
  if (go && X.direct && lane == 0) X.out->ok = 1;   // splitAlign() returned true, rows written by masks_finish
  split_detect(A, X, S, L, go && !X.direct, Ltot, Ltot, lane);
}

__global__ __launch_bounds__(WAVE) void lr_ins_kernel(SplitArgs A, LrInsArgs R) {
  __shared__ PostLR L;
  __shared__ MyersLds<MYERS_NW> ML;
  const int lane = threadIdx.x;
  myers_lut_init(ML.lut, lane);
  __syncthreads();
  uint8_t* ws = R.ws + (size_t)blockIdx.x * R.ws_stride;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    const int j = A.work_list[w];
    if (j < 0) continue;
    process_lr_ins(A, R, j, L, ML, ws, lane);
  }
}

}  // namespace dh
