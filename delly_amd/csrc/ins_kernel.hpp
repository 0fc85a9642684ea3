// ins_kernel.hpp -- gfx950 device code for insertion junctions (svt 4):
//   alignConsensus -> _consRefAlignment -> splitAlign   (src/split.h:480-552, 644-666)
//   edlibAlign HW / SHW with PATH                        (src/edlib.cpp:139-300, vendored edlib)
//   editDistanceVec, glueAlignment                       (src/split.h:377-477)
//   followed by the shared _findSplit ... alleles stage  (split_main.hpp: split_detect)
//
// edlib computes Myers' bit-vector recurrences inside an Ukkonen band with a doubling k; all
// of its observable outputs (edit distance, first end location, HW start location, the op
// string of obtainAlignmentTraceback) are functions of the exact unit-cost DP matrix, so the
// device computes that matrix directly, one junction per 64-lane wavefront:
//   rows    = target letters (the consensus, <= 319: K <= 5 rows per lane),
//   columns = query letters (pieces of the reference window), one column per lane of skew,
//   row hand-off with DPP wave shifts -- the same systolic layout as the longNeedle kernels.
// Where a path is needed the pass stores the op edlib's traceback would pick in each cell
// (2 bits: MATCH / INSERT / DELETE / MISMATCH, preference INSERT > DELETE > diagonal,
// edlib.cpp:1018-1125) and the traceback walks those codes.
//
// Hirschberg mode of edlib (alignment data >= 1 MiB, edlib.cpp:1188) never triggers for
// |query| <= 2048, |target| <= 319.
#pragma once
#include "split_main.hpp"

namespace dh {

constexpr int POSBIG = 1 << 28;
enum : int { ED_MATCH = 0, ED_INSERT = 1, ED_DELETE = 2, ED_MISMATCH = 3 };  // EDLIB_EDOP_*

struct __attribute__((aligned(16))) InsLds {
  StrLds s;   // cons; rcons := reverseComplement(cs); ref = svRefStr; rref = reverseComplement(svRefStr)
  PostLds p;  // trF / trR: op strings (reverse order) of cigarLeft / cigarRight; column masks
  uint16_t distF[NMAX], distR[NMAX];  // editDistanceVec outputs
};

struct EdKeys {
  unsigned kf;  // min over rows of (E[r][qlen] << 12) | r          -> first optimal end
  unsigned kl;  // min over rows of (E[r][qlen] << 12) | (4095 - r) -> last optimal end
  int cnt;      // number of rows attaining the minimum (edlib's numLocations)
};

// One DP pass.  Row r (slot r = lane*K + i, 0..tlen) is target prefix length r, column c is
// query prefix length c:  E[r][0] = hw ? 0 : r,  E[0][c] = c,
//   E[r][c] = min(E[r-1][c-1] + (t[r-1] != q[c-1]), E[r-1][c] + 1, E[r][c-1] + 1).
// t[i] = tp[i*tstep], q[i] = qp[i*qstep] (LDS; step -1 gives the reversed strings of
// edlib.cpp:210-212).  DIRS: store edlib's traceback choice per cell, 16 columns per dword,
// dirs[((t >> 4)*K + i)*64 + lane] with t = c + lane - 1.  Returns the two location keys over
// rows r0..tlen of the last column.
template <int K, bool DIRS>
__device__ __forceinline__ EdKeys pass_ed(const uint8_t* tp, int tstep, int tlen, const uint8_t* qp, int qstep,
                                          int qlen, int hw, int r0, uint32_t* dirs, int lane) {
  int a[K], h[K], colq[K];
  uint32_t acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int s = lane * K + i;
    a[i] = (s >= 1 && s <= tlen) ? (int)tp[(s - 1) * tstep] : NOMATCH;
    h[i] = hw ? 0 : s;
    colq[i] = h[i];
    acc[i] = 0;
  }
  const int T = qlen + tlen / K;
  const int nblk = (T + 15) >> 4;
  int upPrev = POSBIG;
  int b = NOMATCH;
  int c = -lane;
  for (int blk = 0; blk < nblk; ++blk) {
    const int ci = blk * 16 + (lane & 15);
    const int chunk = (ci < qlen) ? (int)qp[ci * qstep] : NOMATCH;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      b = dpp_from_prev(b, newc);
      const int recv = dpp_from_prev(h[K - 1], POSBIG);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)qlen) {
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const int x = diag + ((a[i] != b) ? 1 : 0);
          const int y = up + 1;     // consumes a target letter only : DELETE
          const int z = h[i] + 1;   // consumes a query letter only  : INSERT
          const int nv = min(min(x, y), z);
          if (DIRS) {
            const uint32_t code = (z == nv) ? (uint32_t)ED_INSERT
                                            : ((y == nv) ? (uint32_t)ED_DELETE
                                                         : ((diag == nv) ? (uint32_t)ED_MATCH : (uint32_t)ED_MISMATCH));
            acc[i] |= code << (2 * f);
          }
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
    }
    if (DIRS) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        dirs[((size_t)blk * K + i) * WAVE + lane] = acc[i];
        acc[i] = 0;
      }
    }
  }
  unsigned kf = 0xffffffffu, kl = 0xffffffffu;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int r = lane * K + i;
    if (r >= r0 && r <= tlen) {
      kf = min(kf, ((unsigned)colq[i] << 12) | (unsigned)r);
      kl = min(kl, ((unsigned)colq[i] << 12) | (unsigned)(4095 - r));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    kf = min(kf, (unsigned)__shfl_xor((int)kf, o));
    kl = min(kl, (unsigned)__shfl_xor((int)kl, o));
  }
  EdKeys k;
  k.kf = (unsigned)rfl((int)kf);
  k.kl = (unsigned)rfl((int)kl);
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int r = lane * K + i;
    cnt += __popcll(__ballot(r >= r0 && r <= tlen && (unsigned)colq[i] == (k.kf >> 12)));
  }
  k.cnt = cnt;
  return k;
}

__device__ __forceinline__ int ed_rows_per_lane(int tlen) { return (tlen + 1 + WAVE - 1) / WAVE; }

// rows-per-lane dispatch; two instances in the kernel (with / without direction output)
template <bool DIRS>
__device__ __noinline__ EdKeys ed_pass(const uint8_t* tp, int tstep, int tlen, const uint8_t* qp, int qstep, int qlen,
                                       int hw, int r0, uint32_t* dirs, int lane) {
  const int kd = ed_rows_per_lane(tlen);
  EdKeys k;
  k.kf = k.kl = 0xffffffffu;
  k.cnt = 0;
  if (kd <= 1) k = pass_ed<1, DIRS>(tp, tstep, tlen, qp, qstep, qlen, hw, r0, dirs, lane);
  else if (kd == 2) k = pass_ed<2, DIRS>(tp, tstep, tlen, qp, qstep, qlen, hw, r0, dirs, lane);
  else if (kd == 3) k = pass_ed<3, DIRS>(tp, tstep, tlen, qp, qstep, qlen, hw, r0, dirs, lane);
  else if (kd == 4) k = pass_ed<4, DIRS>(tp, tstep, tlen, qp, qstep, qlen, hw, r0, dirs, lane);
  else k = pass_ed<5, DIRS>(tp, tstep, tlen, qp, qstep, qlen, hw, r0, dirs, lane);
  if (DIRS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return k;
}

// obtainAlignmentTraceback (edlib.cpp:943-1143) on the stored op codes, from cell (rr, cc);
// windowed like traceback<K> of split_kernel.hpp.  tr[] receives the ops in REVERSE order
// (last op first), including the straight run at the border.  Returns the op count.
template <int K>
__device__ __forceinline__ int traceback_ed_k(const uint32_t* dirs, int rr, int cc, uint8_t* tr, int lane) {
  GeoK<K> G{dirs};
  const int tl = traceback_runs<true>(G, rr, cc, tr, lane);
  // border runs: target exhausted -> INSERTs (edlib.cpp:1057-1062,1079-1084), query exhausted ->
  // DELETEs (:1027-1031,1087-1091)
  const int tail = (rr > 0) ? rr : cc;
  const uint8_t op = (rr > 0) ? (uint8_t)ED_DELETE : (uint8_t)ED_INSERT;
  for (int k = lane; k < tail; k += WAVE) tr[tl + k] = op;
  return tl + tail;
}

__device__ __noinline__ int traceback_ed(const uint32_t* dirs, int kd, int rr, int cc, uint8_t* tr, int lane) {
  rr = rfl(rr);
  cc = rfl(cc);
  kd = rfl(kd);
  int n;
  if (kd <= 1) n = traceback_ed_k<1>(dirs, rr, cc, tr, lane);
  else if (kd == 2) n = traceback_ed_k<2>(dirs, rr, cc, tr, lane);
  else if (kd == 3) n = traceback_ed_k<3>(dirs, rr, cc, tr, lane);
  else if (kd == 4) n = traceback_ed_k<4>(dirs, rr, cc, tr, lane);
  else n = traceback_ed_k<5>(dirs, rr, cc, tr, lane);
  __syncthreads();
  return rfl(n);
}

struct EdRes {
  int ed;        // editDistance
  int endLoc;    // endLocations[0]
  int startLoc;  // startLocations[0]
  int nops;      // alignmentLength (ops in tr[], reverse order)
  int nloc;      // numLocations
};

// number of the first admissible row of the last column: "position -1" (no target letter
// consumed) is observed by edlib only through the W padding rows of the last 64-row block,
// i.e. when |query| is not a multiple of 64 (edlib.cpp:653-691)
__device__ __forceinline__ int ed_first_row(int qn) { return ((qn & 63) != 0) ? 0 : 1; }

__device__ __forceinline__ int fill_inserts(uint8_t* tr, int qn, int lane) {
  for (int k = lane; k < qn; k += WAVE) tr[k] = (uint8_t)ED_INSERT;
  __syncthreads();
  return qn;
}

// edlibAlign(Q, T, HW): distance + first end location; LOC adds the start location
// (edlib.cpp:205-249: SHW of the reversed query over the reversed target prefix, LAST optimal
// end); PATH adds the NW traceback over T[start..end] (edlib.cpp:262-276, 1163-1201).
// Requires qn >= 1, tn >= 1.
__device__ __forceinline__ EdRes ed_hw(const uint8_t* T, int tn, const uint8_t* Q, int qn, bool loc, bool path,
                                       uint32_t* dirs, uint8_t* tr, int lane) {
  EdRes o;
  const int r0 = ed_first_row(qn);
  EdKeys k = ed_pass<false>(T, 1, tn, Q, 1, qn, 1, r0, dirs, lane);
  o.ed = (int)(k.kf >> 12);
  o.endLoc = (int)(k.kf & 4095u) - 1;
  o.startLoc = 0;
  o.nops = 0;
  o.nloc = k.cnt;
  if (!loc) return o;
  if (o.endLoc == -1) {  // edlib.cpp:222-235
    if (path) o.nops = fill_inserts(tr, qn, lane);
    return o;
  }
  const int tl = o.endLoc + 1;
  EdKeys k2 = ed_pass<false>(T + o.endLoc, -1, tl, Q + (qn - 1), -1, qn, 0, r0, dirs, lane);
  const int lastj = 4095 - (int)(k2.kl & 4095u);
  o.startLoc = o.endLoc - (lastj - 1);
  if (!path) return o;
  const int tl2 = o.endLoc - o.startLoc + 1;
  if (tl2 <= 0) {  // obtainAlignment, edlib.cpp:1169-1176
    o.nops = fill_inserts(tr, qn, lane);
    return o;
  }
  (void)ed_pass<true>(T + o.startLoc, 1, tl2, Q, 1, qn, 0, 0, dirs, lane);
  o.nops = traceback_ed(dirs, ed_rows_per_lane(tl2), tl2, qn, tr, lane);
  return o;
}

// edlibAlign(Q, T, SHW, PATH).  The NW matrix over T[0..end] that edlib recomputes for the
// path is the top part of the SHW matrix itself (same borders), so one pass serves both.
__device__ __forceinline__ EdRes ed_shw(const uint8_t* T, int tn, const uint8_t* Q, int qn, uint32_t* dirs, uint8_t* tr,
                                        int lane) {
  EdRes o;
  const int r0 = ed_first_row(qn);
  EdKeys k = ed_pass<true>(T, 1, tn, Q, 1, qn, 0, r0, dirs, lane);
  o.ed = (int)(k.kf >> 12);
  o.endLoc = (int)(k.kf & 4095u) - 1;
  o.startLoc = 0;
  o.nloc = k.cnt;
  if (o.endLoc == -1) o.nops = fill_inserts(tr, qn, lane);
  else o.nops = traceback_ed(dirs, ed_rows_per_lane(tn), o.endLoc + 1, qn, tr, lane);
  return o;
}

// editDistanceVec (split.h:377-405): dist[q] = edits among the ops up to the one that consumes
// query letter q.  tr[] holds the ops in reverse order.
__device__ __forceinline__ void edit_distance_vec(const uint8_t* tr, int nops, uint16_t* dist, int lane) {
  int qbase = 0, ebase = 0;
  const unsigned long long le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  for (int k = 0; k < nops; k += WAVE) {
    const int idx = k + lane;
    const int op = (idx < nops) ? (int)tr[nops - 1 - idx] : ED_DELETE;
    const bool isq = (idx < nops) && (op != ED_DELETE);
    const bool ise = (idx < nops) && (op != ED_MATCH);
    const unsigned long long bq = __ballot(isq), be = __ballot(ise);
    if (isq) dist[qbase + __popcll(bq & le) - 1] = (uint16_t)(ebase + __popcll(be & le));
    qbase += __popcll(bq);
    ebase += __popcll(be);
  }
  __syncthreads();
}

// appends the forward op string (tr[] reversed) to the column masks: consensus (target)
// letter unless INSERT, reference (query) letter unless DELETE
__device__ __forceinline__ int mask_append_ops(PostLds& L, int pos, const uint8_t* tr, int nops, int lane) {
  for (int k = 0; k < nops; k += WAVE) {
    const int idx = k + lane;
    const int op = (idx < nops) ? (int)tr[nops - 1 - idx] : 0;
    const unsigned long long v = __ballot(idx < nops && op != ED_INSERT);
    const unsigned long long r = __ballot(idx < nops && op != ED_DELETE);
    const int cnt = min(WAVE, nops - k);
    mask_append(L, pos, cnt, v, r, lane);
    pos += cnt;
  }
  return pos;
}
__device__ __forceinline__ int mask_append_run(PostLds& L, int pos, int cnt, unsigned long long v, unsigned long long r,
                                               int lane) {
  for (int k = 0; k < cnt; k += WAVE) {
    mask_append(L, pos, min(WAVE, cnt - k), v, r, lane);
    pos += min(WAVE, cnt - k);
  }
  return pos;
}

// alignConsensus for one svt 4 junction
__device__ void process_ins(const SplitArgs& A, int j, InsLds& L, uint32_t* scratch, int lane) {
  JCtx X;
  junction_setup<KMAX, true, StrLds, true>(A, j, L.s, X, lane);
  const int m = X.m, n = X.n;
  bool go = X.go;
  if (go && m < 1) {  // empty consensus: edlib's zero-length special cases are not mirrored
    if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
    go = false;
  }
  int Ltot = 0;
  if (go) {
    const uint8_t* cons = L.s.cons;
    const uint8_t* ref = L.s.ref;
    // --- splitAlign, split.h:483-492: where does the reference window sit inside the consensus?
    const EdRes pre = ed_hw(cons, m, ref, n / 3, true, false, scratch, L.p.trF, lane);
    const uint32_t csStart = (uint32_t)pre.startLoc;  // infixStart == startLocations[0]
    const int so = (int)((2ull * (uint64_t)n) / 3ull);
    const EdRes suf = ed_hw(cons, m, ref + so, n - so, false, false, scratch, L.p.trF, lane);
    const uint32_t csEnd = (uint32_t)suf.endLoc;      // infixEnd
    if (lane == 0) {
      X.out->score_unsplit = (int32_t)csStart;
      X.out->score_best = (int32_t)csEnd;
    }
    if (csStart >= csEnd) go = false;
    int bestJoin = 0;
    if (go) {
      // cs = cons.substr(csStart, csEnd - csStart)
      uint32_t cslu = csEnd - csStart;
      if (cslu > (uint32_t)m - csStart) cslu = (uint32_t)m - csStart;
      const int csl = (int)cslu;
      const uint8_t* cs = cons + csStart;
      if (csl == 0) {  // edlib's empty-target special case returns no alignment: all distances 0
        for (int i = lane; i < n; i += WAVE) { L.distF[i] = 0; L.distR[i] = 0; }
        __syncthreads();
      } else {
        // split.h:495-511: prefix distances forward, suffix distances on the reverse complements
        const EdRes f = ed_shw(cs, csl, ref, n, scratch, L.p.trF, lane);
        edit_distance_vec(L.p.trF, f.nops, L.distF, lane);
        for (int i = lane; i < csl; i += WAVE) L.s.rcons[i] = rc_at(cs, csl, i);
        __syncthreads();
        const EdRes r = ed_shw(L.s.rcons, csl, L.s.rref, n, scratch, L.p.trF, lane);
        edit_distance_vec(L.p.trF, r.nops, L.distR, lane);
      }
      // best join, split.h:513-517: first minimum of distFwd[i] + distRev[n-i-2], i = 0..n-2
      unsigned key = 0xffffffffu;
      for (int i = lane; i <= n - 2; i += WAVE) {
        const unsigned sum = (unsigned)L.distF[i] + (unsigned)L.distR[n - i - 2];
        key = min(key, (sum << 12) | (unsigned)i);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
      bestJoin = rfl((int)(key & 4095u));
      if (lane == 0) X.out->cons_left = bestJoin;
    }
    EdRes le, ri;
    le.nops = ri.nops = 0;
    uint32_t gaplen = 0, missingStart = 0, missingEnd = 0;
    if (go) {
      // split.h:519-530: infix alignments of the two reference halves
      le = ed_hw(cons, m, ref, bestJoin + 1, true, true, scratch, L.p.trF, lane);
      ri = ed_hw(cons, m, ref + bestJoin + 1, n - bestJoin - 1, true, true, scratch, L.p.trR, lane);
      const uint32_t leftEnd = (uint32_t)le.endLoc, rightStart = (uint32_t)ri.startLoc;
      if (lane == 0) {
        X.out->ref_left = (int32_t)leftEnd;
        X.out->ref_right = (int32_t)rightStart;
      }
      if (leftEnd + 15u >= rightStart) go = false;  // split.h:532
      gaplen = rightStart - leftEnd - 1u;
      missingStart = (uint32_t)le.startLoc;         // glueAlignment, split.h:413-421
      missingEnd = (uint32_t)ri.endLoc;
      if (missingEnd < (uint32_t)m) missingEnd = (uint32_t)m - missingEnd - 1u;
      if (go) {
        const uint64_t total = (uint64_t)missingStart + (uint64_t)le.nops + gaplen + (uint64_t)ri.nops + missingEnd;
        if (total > (uint64_t)(MASKW * 64) || total > (uint64_t)TRACE_CAP) {
          if (lane == 0) X.out->status = DELLYHIP_E_LIMIT;
          go = false;
        }
      }
    }
    go = rfl((int)go) != 0;
    if (go) {
      // glueAlignment (split.h:407-477) + row swap (:548-552) as column masks:
      // [consensus only x missingStart][left ops][consensus only x gaplen][right ops][consensus only x missingEnd]
      for (int w = lane; w < MASKW; w += WAVE) {
        L.p.mV[w] = 0;
        L.p.mR[w] = 0;
        L.p.mE[w] = 0;
      }
      __syncthreads();
      int pos = 0;
      pos = mask_append_run(L.p, pos, (int)missingStart, ~0ull, 0ull, lane);
      pos = mask_append_ops(L.p, pos, L.p.trF, le.nops, lane);
      pos = mask_append_run(L.p, pos, (int)gaplen, ~0ull, 0ull, lane);
      pos = mask_append_ops(L.p, pos, L.p.trR, ri.nops, lane);
      pos = mask_append_run(L.p, pos, (int)missingEnd, ~0ull, 0ull, lane);
      Ltot = pos;
      masks_finish(A, X, L.s, L.p, Ltot, Ltot, lane);
    }
  }
  X.go = go;
  if (go && X.direct && lane == 0) X.out->ok = 1;   // dellyhip_split_align: splitAlign() returned true, rows written by masks_finish
  split_detect(A, X, L.s, L.p, go && !X.direct, Ltot, Ltot, lane);
}

// ---- single edlibAlign call (parity tests; edlib.h:242-246) ----------------------------
struct EdArgs {
  const uint8_t* q;   // query, qn <= NMAX
  const uint8_t* t;   // target, tn <= MMAX
  int32_t qn, tn;
  int32_t mode;       // EdlibAlignMode: 0 NW, 1 SHW, 2 HW
  int32_t task;       // EdlibAlignTask: 0 DISTANCE, 1 LOC, 2 PATH
  int32_t* out;       // {editDistance, numLocations, endLocations[0], startLocations[0] (-2: not computed), alignmentLength}
  uint8_t* ops;       // alignment (EDLIB_EDOP_*), forward order
  uint32_t* scratch;
};

__global__ __launch_bounds__(WAVE) void edlib_single_kernel(EdArgs a) {
  __shared__ uint8_t q[NMAX];
  __shared__ uint8_t t[MMAX + 1];
  __shared__ uint8_t tr[TRACE_CAP];
  const int lane = threadIdx.x;
  const int qn = a.qn, tn = a.tn;
  for (int i = lane; i < qn; i += WAVE) q[i] = a.q[i];
  for (int i = lane; i < tn; i += WAVE) t[i] = a.t[i];
  __syncthreads();
  EdRes o;
  const bool loc = a.task >= 1, path = a.task >= 2;
  if (a.mode == 2) {
    o = ed_hw(t, tn, q, qn, loc, path, a.scratch, tr, lane);
  } else if (a.mode == 1) {
    if (path) o = ed_shw(t, tn, q, qn, a.scratch, tr, lane);
    else {
      const EdKeys k = ed_pass<false>(t, 1, tn, q, 1, qn, 0, ed_first_row(qn), a.scratch, lane);
      o.ed = (int)(k.kf >> 12);
      o.endLoc = (int)(k.kf & 4095u) - 1;
      o.startLoc = 0;
      o.nops = 0;
      o.nloc = k.cnt;
    }
  } else {  // NW: the single admissible end is the last row
    EdKeys k;
    if (path) k = ed_pass<true>(t, 1, tn, q, 1, qn, 0, tn, a.scratch, lane);
    else k = ed_pass<false>(t, 1, tn, q, 1, qn, 0, tn, a.scratch, lane);
    o.ed = (int)(k.kf >> 12);
    o.endLoc = tn - 1;
    o.startLoc = 0;
    o.nloc = 1;
    o.nops = path ? traceback_ed(a.scratch, ed_rows_per_lane(tn), tn, qn, tr, lane) : 0;
  }
  __syncthreads();
  for (int i = lane; i < o.nops; i += WAVE) a.ops[i] = tr[o.nops - 1 - i];
  if (lane == 0) {
    a.out[0] = o.ed;
    a.out[1] = o.nloc;
    a.out[2] = o.endLoc;
    a.out[3] = loc ? o.startLoc : -2;
    a.out[4] = o.nops;
  }
}

__global__ __launch_bounds__(WAVE) void ins_kernel(SplitArgs A) {
  __shared__ InsLds L;
  const int lane = threadIdx.x;
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    const int j = A.work_list[w];
    if (j < 0) continue;
    process_ins(A, j, L, scratch, lane);
  }
}

}  // namespace dh
