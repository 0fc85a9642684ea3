// split_main.hpp -- per-junction driver of the split-alignment kernel:
// alignConsensus() of src/split.h:644-666 for svt != 4, one junction per wave.
#pragma once
#include "split_kernel.hpp"

namespace dh {

// wave-wide max of a signed 64-bit key
__device__ __forceinline__ long long wave_max64(long long v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    int lo = __shfl_xor((int)(v & 0xffffffffll), o);
    int hi = __shfl_xor((int)(v >> 32), o);
    long long w = ((long long)hi << 32) | (unsigned int)lo;
    v = (w > v) ? w : v;
  }
  return v;
}

// fills dst[0..len) of the window string from one segment (parallel over lanes)
__device__ __forceinline__ void fill_segment(uint8_t* dst, const Seg& sg, int lane) {
  for (int i = lane; i < sg.len; i += WAVE) {
    uint8_t fwd = upc(sg.base[sg.beg + i]);
    if (!sg.rc) dst[i] = fwd;
    else {
      uint8_t r = comp_acgtn(upc(sg.base[sg.beg + (sg.len - 1 - i)]));
      dst[i] = r ? r : fwd;  // split.h:83-90: default keeps the un-reversed byte
    }
  }
}

__device__ __forceinline__ bool is_tra(int svt) { return svt >= 5 && svt < 9; }

template <int K>
__device__ void process_junction(const SplitArgs& A, int j, WaveLds& L, uint32_t* scratch, int lane) {
  const dellyhip_junction J = A.junc[j];
  const dellyhip_params& P = A.p;
  dellyhip_result R;
  // zero-initialise (uniform)
  {
    int* rp = reinterpret_cast<int*>(&R);
#pragma unroll
    for (unsigned q = 0; q < sizeof(R) / 4; ++q) rp[q] = 0;
  }
  R.svid = J.svid;
  R.sv_start = J.sv_start;
  R.sv_end = J.sv_end;
  R.ins_len = J.ins_len;
  R.score_unsplit = R.score_best = R.cons_left = R.ref_left = R.ref_right = -1;
  R.matches = R.mismatches = -1;
  uint8_t* ob = A.out_blob + (size_t)j * A.out_stride;
  const uint64_t ob_off = (uint64_t)j * A.out_stride;
  const int m = A.cons_len[j];
  const uint8_t* cons_g = A.cons_base + A.cons_off[j];
  R.cons_len = m;
  R.cons_off = ob_off;
  R.sr_support = A.res[j].sr_support;  // written by the MSA stage (0 otherwise)
  dellyhip_result* out = &A.res[j];

  bool go = true;
  const int prior = A.res[j].status;  // set by the MSA stage (kernel limit exceeded there)
  if (prior) {
    R.status = prior;
    R.cons_len = 0;
    go = false;
  } else if (m < 0 || m > MMAX || m + 1 > WAVE * K) {
    R.status = DELLYHIP_E_LIMIT;
    R.cons_len = 0;
    go = false;
  }
  if (go) {
    for (int i = lane; i < m; i += WAVE) {
      uint8_t ch = cons_g[i];
      L.cons[i] = ch;
      if (cons_g != ob) ob[i] = ch;
    }
  }
  const bool direct = (A.ref_base != nullptr);
  if (go && !direct && J.svt == 4) {  // splitAlign/edlib path: not in this kernel
    R.status = DELLYHIP_E_LIMIT;
    go = false;
  }
  if (go && !direct && m < 2 * P.minimum_flank_size + J.ins_len) go = false;  // split.h:647

  // ---- _initBreakpoint (tags.h:151-172) + _getSVRef segments (split.h:70-163)
  int sBeg = 0, sEnd = 0, eBeg = 0, eEnd = 0;
  Seg seg[3];
  int nseg = 0, n = 0;
  if (go && direct) {
    n = A.ref_len[j];
    R.ref_len = n;
    if (n > NMAX || n < 0) { R.status = DELLYHIP_E_LIMIT; go = false; }
    else {
      const uint8_t* rg = A.ref_base + A.ref_off[j];
      for (int i = lane; i < n; i += WAVE) L.ref[i] = rg[i];
    }
  } else if (go) {
    const int boundary = m;
    const int svS = J.sv_start, svE = J.sv_end;
    const int len1 = (int)(uint32_t)A.chr_len[J.chr], len2 = (int)(uint32_t)A.chr_len[J.chr2];
    const uint8_t* c1 = A.chr_seq[J.chr];
    const uint8_t* c2 = A.chr_seq[J.chr2];
    if (is_tra(J.svt)) {
      sBeg = max(0, svS - boundary);
      sEnd = min(len1, svS + boundary);
      eBeg = max(0, svE - boundary);
      eEnd = min(len2, svE + boundary);
      int ct = J.svt - 5;
      Seg mainS{c1, sBeg, max(0, sEnd - sBeg), ct == 1};
      if (J.chr != J.chr2) {
        Seg part1{c2, eBeg, max(0, eEnd - eBeg), ct == 0};
        if (ct == 3) { seg[0] = part1; seg[1] = mainS; }
        else { seg[0] = mainS; seg[1] = part1; }
        nseg = 2;
      } else {
        seg[0] = mainS;
        nseg = 1;
      }
    } else {
      int mid = (svS + svE) / 2;
      sBeg = max(0, svS - boundary);
      sEnd = min(svS + boundary, mid);
      eBeg = max(mid + 1, svE - boundary);
      eEnd = min(len2, svE + boundary);
      if (J.svt == 2) {
        if (svE - svS <= P.indelsize) { seg[0] = Seg{c1, sBeg, max(0, eEnd - sBeg), 0}; nseg = 1; }
        else { seg[0] = Seg{c1, sBeg, max(0, sEnd - sBeg), 0}; seg[1] = Seg{c1, eBeg, max(0, eEnd - eBeg), 0}; nseg = 2; }
      } else if (J.svt == 3) {
        seg[0] = Seg{c1, eBeg, max(0, eEnd - eBeg), 0};
        seg[1] = Seg{c1, sBeg, max(0, sEnd - sBeg), 0};
        nseg = 2;
      } else if (J.svt == 0) {
        seg[0] = Seg{c1, sBeg, max(0, sEnd - sBeg), 0};
        if (svE - svS > P.min_cons_window) { seg[1] = Seg{c1, eBeg, max(0, eEnd - eBeg), 1}; nseg = 2; }
        else { seg[1] = Seg{c1, svS, max(0, eEnd - svS), 1}; seg[2] = Seg{c1, svE, max(0, eEnd - svE), 0}; nseg = 3; }
      } else if (J.svt == 1) {
        if (svE - svS > P.min_cons_window) {
          seg[0] = Seg{c1, sBeg, max(0, sEnd - sBeg), 1};
          seg[1] = Seg{c1, eBeg, max(0, eEnd - eBeg), 0};
          nseg = 2;
        } else {
          seg[0] = Seg{c1, sBeg, max(0, svS - sBeg), 0};
          seg[1] = Seg{c1, sBeg, max(0, svE - sBeg), 1};
          seg[2] = Seg{c1, eBeg, max(0, eEnd - eBeg), 0};
          nseg = 3;
        }
      } else {
        go = false;  // unknown svt: _getSVRef returns "" -> longNeedle on empty ref finds nothing
      }
    }
    for (int q = 0; q < nseg; ++q) n += seg[q].len;
    R.ref_len = n;
    if (go && n > NMAX) {
      R.status = DELLYHIP_E_LIMIT;
      go = false;
    }
  }
  if (go) {
    int o = 0;
    for (int q = 0; q < nseg; ++q) {
      fill_segment(L.ref + o, seg[q], lane);
      o += seg[q].len;
    }
  }
  __syncthreads();
  if (go) {
    // reverseComplement(s1), reverseComplement(s2): util.h:549-563
    for (int i = lane; i < m; i += WAVE) {
      uint8_t r = comp_acgtn(upc(L.cons[m - 1 - i]));
      L.rcons[i] = r ? r : L.cons[i];
    }
    for (int i = lane; i < n; i += WAVE) {
      uint8_t r = comp_acgtn(upc(L.ref[n - 1 - i]));
      L.rref[i] = r ? r : L.ref[i];
    }
  }
  __syncthreads();

  // ---- longNeedle: R-pass, M-pass + join (needle.h:52-123)
  int consLeft = 0, refLeft = 0, refRight = 0, consRight = 0, best = 0, unsplit = 0;
  if (go) {
    int hfin[K], brfin[K], bestkey[K];
    int hrow_m;
    pass_R<K>(L, m, n, scratch, lane, hfin, brfin);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pass_M<K>(L, m, n, scratch, lane, brfin, bestkey, hrow_m);
    // rev[m][n]: slot m of the R-pass
    int revmn = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if (lane * K + i == m) revmn = hfin[i];
    revmn = __shfl(revmn, m / K) - m;
    unsplit = __shfl(hrow_m, 0);
    unsplit = (unsplit >> SCALE_SHIFT) - m;
    // first (row, col) in row-major order that attains the maximum (needle.h:107-115):
    // max sum', then max slot (= min row), then max cinv (= min col)
    long long key = (long long)0x8000000000000000ll;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      int s = lane * K + i;
      if (s <= m) {
        long long kk = ((long long)(bestkey[i] >> SCALE_SHIFT) << 32) | ((long long)s << 12) | (bestkey[i] & 4095);
        key = kk > key ? kk : key;
      }
    }
    key = wave_max64(key);
    best = (int)(key >> 32) - m;  // sum' = sum + m
    int sstar = (int)((key >> 12) & 0xfffff);
    refLeft = 4095 - (int)(key & 4095);
    consRight = sstar;
    consLeft = m - sstar;
    R.score_unsplit = unsplit;
    if (unsplit != revmn) go = false;  // needle.h:83-85
    else {
      if (best <= unsplit) {  // no improving join: consLeft = refLeft = 0, bestScore = mat[m][n]
        best = unsplit;
        consLeft = 0;
        refLeft = 0;
        consRight = m;
      }
      // refRight: last t in [0, n-refLeft] with rev[consRight][t] == running max there (needle.h:119-123)
      {
        int ls = consRight / K, is = consRight - ls * K;
        int X = n - refLeft;
        refRight = 0;
        int t = X + ls - 1;  // zero-based producer step of column X
        bool hit = false;
        while (!hit && X >= 1) {
          uint32_t w = ld_scratch(&scratch[((size_t)(t >> 4) * K + is) * WAVE + ls]);
          w = (uint32_t)rfl((int)w);
          int f = t & 15;
          // fields f, f-1, ... 0 of this word cover columns X, X-1, ...
          uint32_t keep = (f == 15) ? 0xffffffffu : ((1u << (2 * f + 2)) - 1u);
          uint32_t x = w & keep;
          int ncols_here = min(f + 1, X);  // columns X .. X-ncols_here+1
          if (ncols_here < f + 1) x &= ~((1u << (2 * (f + 1 - ncols_here))) - 1u);  // columns < 1 do not exist
          if (x) {
            int top = (31 - __builtin_clz(x)) >> 1;  // highest non-zero field
            refRight = X - (f - top);
            hit = true;
          } else {
            X -= ncols_here;
            t -= ncols_here;
          }
        }
      }
      R.score_best = best;
      R.cons_left = consLeft;
      R.ref_left = refLeft;
      R.ref_right = refRight;
      if (best == unsplit) go = false;  // needle.h:152
    }
  }

  // ---- tracebacks (needle.h:154-194) on recomputed direction codes
  int nF = 0, tvF = 0, thF = 0, nR = 0, tvR = 0, thR = 0;
  if (go) {
    if (consLeft > 0 && refLeft > 0) {
      pass_dir<K>(L.cons, L.ref, m, consLeft, refLeft, scratch, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      nF = traceback<K>(scratch, consLeft, refLeft, L.trF, lane, tvF, thF);
    } else {
      tvF = consLeft;
      thF = (consLeft > 0) ? 0 : refLeft;
      if (consLeft > 0 && refLeft == 0) thF = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (consRight > 0 && refRight > 0) {
      pass_dir<K>(L.rcons, L.rref, m, consRight, refRight, scratch, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      nR = traceback<K>(scratch, consRight, refRight, L.trR, lane, tvR, thR);
    } else {
      tvR = consRight;
      thR = (consRight > 0) ? 0 : refRight;
    }
  }
  __syncthreads();

  // ---- alignment as column masks -------------------------------------------------
  // column order (needle.h:196-219): [fwd tail][fwd ops reversed][ref gap][rev ops][rev tail]
  int Ltot = 0, posGap = 0, posC = 0;
  if (go) {
    for (int w = lane; w < MASKW; w += WAVE) {
      L.mV[w] = 0;
      L.mR[w] = 0;
      L.mE[w] = 0;
    }
    __syncthreads();
    int pos = 0;
    // forward tail: thF ref-only columns, or tvF cons-only columns
    for (int k = 0; k < thF; k += 64) { mask_append(L, pos, min(64, thF - k), 0ull, ~0ull, lane); pos += min(64, thF - k); }
    for (int k = 0; k < tvF; k += 64) { mask_append(L, pos, min(64, tvF - k), ~0ull, 0ull, lane); pos += min(64, tvF - k); }
    for (int k = 0; k < nF; k += 64) {
      int idx = k + lane;
      int op = (idx < nF) ? (int)L.trF[nF - 1 - idx] : 0;
      unsigned long long v = __ballot(idx < nF && op != 2);
      unsigned long long r = __ballot(idx < nF && op != 1);
      mask_append(L, pos, min(64, nF - k), v, r, lane);
      pos += min(64, nF - k);
    }
    posGap = pos;
    int gapref = (n - refRight) - refLeft;
    for (int k = 0; k < gapref; k += 64) { mask_append(L, pos, min(64, gapref - k), 0ull, ~0ull, lane); pos += min(64, gapref - k); }
    posC = pos;
    for (int k = 0; k < nR; k += 64) {
      int idx = k + lane;
      int op = (idx < nR) ? (int)L.trR[idx] : 0;
      unsigned long long v = __ballot(idx < nR && op != 2);
      unsigned long long r = __ballot(idx < nR && op != 1);
      mask_append(L, pos, min(64, nR - k), v, r, lane);
      pos += min(64, nR - k);
    }
    for (int k = 0; k < tvR; k += 64) { mask_append(L, pos, min(64, tvR - k), ~0ull, 0ull, lane); pos += min(64, tvR - k); }
    for (int k = 0; k < thR; k += 64) { mask_append(L, pos, min(64, thR - k), 0ull, ~0ull, lane); pos += min(64, thR - k); }
    Ltot = pos;
    __syncthreads();
    // cumulative counts
    if (lane == 0) {
      int cv = 0, cr = 0;
      int nw = (Ltot + 63) >> 6;
      for (int w = 0; w < nw; ++w) {
        L.cumV[w] = cv;
        L.cumR[w] = cr;
        cv += __popcll(L.mV[w]);
        cr += __popcll(L.mR[w]);
      }
      L.cumV[nw] = cv;
      L.cumR[nw] = cr;
    }
    __syncthreads();
    // characters of every column -> equality mask (+ optional alignment output)
    uint8_t* aln = ob + OUT_CONS_CAP + OUT_ALLELE_CAP;
    for (int base = 0; base < Ltot; base += 64) {
      int jcol = base + lane;
      int w = base >> 6;
      unsigned long long mv = L.mV[w], mr = L.mR[w];
      unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      int cv = L.cumV[w] + __popcll(mv & below);
      int cr = L.cumR[w] + __popcll(mr & below);
      bool v = (mv >> lane) & 1ull, r = (mr >> lane) & 1ull;
      uint8_t c0 = '-', c1 = '-';
      if (jcol < Ltot) {
        if (jcol < posC) {
          if (v) c0 = L.cons[cv];
          if (r) c1 = L.ref[cr];
        } else {
          if (v) c0 = outmap(L.rcons[m - 1 - cv]);
          if (r) c1 = outmap(L.rref[n - 1 - cr]);
        }
      }
      // presence is defined on the characters (a '\0' from outmap still counts as present,
      // exactly like align[0][j] != '-')
      unsigned long long e = __ballot(jcol < Ltot && v && r && c0 == c1);
      if (lane == 0) L.mE[w] = e;
      if (A.want_alignment && jcol < Ltot) {
        aln[jcol] = c0;
        aln[Ltot + jcol] = c1;
      }
    }
    __syncthreads();
    if (A.want_alignment) {
      R.aln_off = ob_off + OUT_CONS_CAP + OUT_ALLELE_CAP;
      R.aln_len = Ltot;
    }
  }

  if (go && direct) R.ok = 1;  // longNeedle() returned true
  // ---- _findSplit (split.h:319-375) on the masks (uniform code)
  if (go && !direct) {
    const int svt = J.svt;
    int nw = (Ltot + 63) >> 6;
    int fv = next_set(L.mV, 0ull, 0, Ltot), fr = next_set(L.mR, 0ull, 0, Ltot);
    int J0 = max(fv, fr);  // first column with varIndex > 0 && refIndex > 0
    int cStart = 0, cEnd = 0, rStart = 0, rEnd = 0, gS = 0, gE = 0;
    int closedLen = 0, chosenLen = 0;
    // gap columns: NOT (v & r), from J0 on.  Build on the fly: G = ~(mV & mR)
    int pos = J0;
    while (pos < Ltot) {
      // next gap column
      int a = pos;
      while (a < Ltot) {
        int w = a >> 6, o = a & 63;
        unsigned long long x = (~(L.mV[w] & L.mR[w])) >> o;
        if (x) { a += __builtin_ctzll(x); break; }
        a = (w + 1) << 6;
      }
      if (a >= Ltot) break;
      // end of the run: next non-gap column
      int b1 = a;
      while (b1 < Ltot) {
        int w = b1 >> 6, o = b1 & 63;
        unsigned long long x = (L.mV[w] & L.mR[w]) >> o;
        if (x) { b1 += __builtin_ctzll(x); break; }
        b1 = (w + 1) << 6;
      }
      if (b1 >= Ltot) break;  // trailing run: never closed, never evaluated
      int ra = cnt_before(L.mR, L.cumR, a), rb = cnt_before(L.mR, L.cumR, b1);
      int va = cnt_before(L.mV, L.cumV, a), vb = cnt_before(L.mV, L.cumV, b1);
      int refspan = rb - ra + 1, varspan = vb - va + 1;
      closedLen += b1 - a;
      bool better = (svt == 4) ? (varspan > (cEnd - cStart)) : (refspan > (rEnd - rStart));
      if (better) {
        rStart = ra; rEnd = ra + refspan; cStart = va; cEnd = va + varspan;
        gS = a; gE = b1 - 1;
        chosenLen = b1 - a;
      }
      pos = b1 + 1;
    }
    (void)nw;
    bool ok = rEnd > rStart;
    if (ok) {
      if (svt == 4) ok = ((rEnd - rStart) < 5) && ((cEnd - cStart) > 15);
      else ok = ((cEnd - cStart) < 5) && ((rEnd - rStart) > 15);
    }
    int ma = 0, mm = 0;
    float percId = 0.f;
    if (ok) {
      // _percentIdentity split.h:282-316
      for (int w = 0; w < ((Ltot + 63) >> 6); ++w) {
        unsigned long long both = L.mV[w] & L.mR[w];
        ma += __popcll(both & L.mE[w]);
        mm += __popcll(both & ~L.mE[w]);
      }
      mm += closedLen - chosenLen;
      percId = (float)(uint32_t)ma / (float)(uint32_t)(ma + mm);
      if (percId < P.flank_quality) ok = false;
    }
    int homLeft = 0, homRight = 0;
    if (ok) {
      // _findHomology split.h:262-280 (svt != 4 in this kernel)
      homRight = longest_homology(L.cons, cEnd - 1, 1, m - (cEnd - 1), L.ref, rStart, 1, n - rStart);
      homLeft = longest_homology(L.cons, cStart - 1, -1, min(cStart, m), L.ref, rEnd - 2, -1, min(rEnd - 1, n));
      const int varIndex = m, refIndex = n;
      if ((homLeft + P.minimum_flank_size > cStart) || (varIndex < cEnd + homRight + P.minimum_flank_size)) ok = false;
      if ((homLeft + P.minimum_flank_size > rStart) || (refIndex < rEnd + homRight + P.minimum_flank_size)) ok = false;
    }
    if (ok) {
      R.c_start = cStart; R.c_end = cEnd; R.r_start = rStart; R.r_end = rEnd;
      R.hom_left = homLeft; R.hom_right = homRight;
      R.matches = ma; R.mismatches = mm;
      // _coordTransform split.h:166-244
      uint32_t gs = 0, ge = 0;
      bool ct_ok = true;
      const int svS = J.sv_start, svE = J.sv_end;
      if (is_tra(svt)) {
        int ct = svt - 5;
        int annealed = (ct == 3) ? (eEnd - eBeg) : (sEnd - sBeg);
        if (rStart >= annealed || rEnd < annealed) ct_ok = false;
        else if (ct == 0) { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)((uint64_t)(int64_t)eBeg + ((uint64_t)n - (uint64_t)(int64_t)rEnd) + 1); }
        else if (ct == 1) { gs = (uint32_t)(sBeg + (annealed - rStart) + 1); ge = (uint32_t)(eBeg + (rEnd - annealed)); }
        else if (ct == 2) { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)(eBeg + (rEnd - annealed)); }
        else { gs = (uint32_t)(sBeg + (rEnd - annealed)); ge = (uint32_t)(eBeg + rStart); }
      } else if (svt == 2) {
        if (svE - svS > P.indelsize) {
          int annealed = sEnd - sBeg;
          if (rStart >= annealed || rEnd < annealed) ct_ok = false;
          else { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)(eBeg + (rEnd - annealed)); }
        } else { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)(sBeg + rEnd); }
      } else if (svt == 3) {
        int annealed = eEnd - eBeg;
        if (rStart >= annealed || rEnd < annealed) ct_ok = false;
        else { gs = (uint32_t)(sBeg + (rEnd - annealed)); ge = (uint32_t)(eBeg + rStart); }
      } else if (svt == 0) {
        int annealed = sEnd - sBeg;
        if (rStart >= annealed || rEnd < annealed) ct_ok = false;
        else if (svE - svS > P.min_cons_window) { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)((uint64_t)(int64_t)eBeg + ((uint64_t)n - (uint64_t)(int64_t)rEnd) + 1); }
        else { gs = (uint32_t)(sBeg + rStart); ge = (uint32_t)(eEnd - (rEnd - annealed)); }
      } else if (svt == 1) {
        int annealed = (svE - svS > P.min_cons_window) ? (sEnd - sBeg) : ((svS - sBeg) + (svE - sBeg));
        if (rStart >= annealed || rEnd < annealed) ct_ok = false;
        else { gs = (uint32_t)(sBeg + (annealed - rStart) + 1); ge = (uint32_t)(eBeg + (rEnd - annealed)); }
      }
      if (ct_ok && (is_tra(svt) || gs < ge)) {
        // exact alleles split.h:606-624
        if ((svE - svS <= P.indelsize) && (svt == 2 || svt == 4)) {
          int colA = (cStart >= 1) ? select_bit(L.mV, L.cumV, cStart, Ltot) : Ltot;
          int colB = select_bit(L.mV, L.cumV, cEnd, Ltot);
          if (colA > colB) colA = colB;
          int rA = cnt_before(L.mR, L.cumR, colA), rB = cnt_before(L.mR, L.cumR, colB);
          int vA = cnt_before(L.mV, L.cumV, colA), vB = cnt_before(L.mV, L.cumV, colB);
          int nr = rB - rA, na = vB - vA;
          uint8_t* al = ob + OUT_CONS_CAP;
          if (nr + na + 1 <= OUT_ALLELE_CAP) {
            for (int base = colA & ~63; base < colB; base += 64) {
              int jcol = base + lane;
              int w = base >> 6;
              unsigned long long mv = L.mV[w], mr = L.mR[w];
              unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
              int cv = L.cumV[w] + __popcll(mv & below);
              int cr = L.cumR[w] + __popcll(mr & below);
              bool v = (mv >> lane) & 1ull, r = (mr >> lane) & 1ull;
              if (jcol >= colA && jcol < colB) {
                if (v) al[nr + 1 + (cv - vA)] = (jcol < posC) ? L.cons[cv] : outmap(L.rcons[m - 1 - cv]);
                if (r) al[cr - rA] = (jcol < posC) ? L.ref[cr] : outmap(L.rref[n - 1 - cr]);
              }
            }
            if (lane == 0) al[nr] = ',';
            R.allele_off = ob_off + OUT_CONS_CAP;
            R.allele_len = nr + na + 1;
          } else {
            R.status = DELLYHIP_E_LIMIT;
          }
        }
        R.ok = 1;
        R.sv_start = (int32_t)gs;
        R.sv_end = (int32_t)ge;
        R.sr_align_quality = percId;
        R.ins_len = cEnd - cStart - 1;
        R.cons_bp = cStart;
        R.hom_len = max(0, homLeft + homRight - 2);
        R.ci_wiggle = max(homLeft, homRight);
      }
    }
    (void)gS; (void)gE; (void)posGap;
  }
  if (lane == 0) *out = R;
  __syncthreads();
}

template <int K>
__global__ __launch_bounds__(WAVE) void split_align_kernel(SplitArgs A) {
  __shared__ WaveLds L;
  const int lane = threadIdx.x;
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(A.work_counter, 1);
    w = rfl(w);
    if (w >= A.n_work) break;
    process_junction<K>(A, A.work_list[w], L, scratch, lane);
  }
}

}  // namespace dh
