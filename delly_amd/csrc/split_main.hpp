// split_main.hpp -- per-junction stages of the split-alignment kernels:
// alignConsensus() of src/split.h:644-666 for svt != 4.
//   junction_setup   : result defaults, consensus, _initBreakpoint/_getSVRef window, reverse complements
//   junction_finish  : score checks, join winner, refRight (needle.h:83-123,152)
//   junction_post    : tracebacks on recomputed direction codes, column masks, _findSplit,
//                      _percentIdentity, _findHomology, _coordTransform, alleles (split.h:166-375,596-637)
// The DP passes between setup and finish are either the one-junction-per-wave passes of
// split_kernel.hpp or the packed two-junctions-per-wave passes of split_pk.hpp.
#pragma once
#include <type_traits>
#include <cstddef>

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

// ---- eight letters per lane (split_sparse_kernel's setup) ----
__device__ __forceinline__ uint64_t ld8u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st8u(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
// upc() on eight bytes
__device__ __forceinline__ uint64_t upc8(uint64_t x) {
  const uint64_t t = x & 0x7f7f7f7f7f7f7f7full;
  const uint64_t a = t + 0x1f1f1f1f1f1f1f1full;      // bit 7 of a byte: t >= 'a'
  const uint64_t b = t + 0x0505050505050505ull;      // bit 7: t > 'z'
  return x - ((a & ~b & ~x & 0x8080808080808080ull) >> 2);
}
// bit 7 of every byte that is NOT one of A, C, G, T, N (0 = all eight are)
__device__ __forceinline__ uint64_t not_acgtn8(uint64_t x) {
  uint64_t any = 0;
  const uint64_t pat[5] = {0x4141414141414141ull, 0x4343434343434343ull, 0x4747474747474747ull, 0x5454545454545454ull, 0x4e4e4e4e4e4e4e4eull};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint64_t z = x ^ pat[i];
    any |= ~((((z & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | z)) & 0x8080808080808080ull;   // byte of z is zero
  }
  return any ^ 0x8080808080808080ull;
}
// comp_acgtn() on eight bytes that are all in A, C, G, T, N: A <-> T is x ^ 0x15, C <-> G is x ^ 0x04; bit 1 is set
// for C, G, N only and bit 3 for N only
__device__ __forceinline__ uint64_t comp8(uint64_t x) {
  const uint64_t one = 0x0101010101010101ull;
  const uint64_t b1 = (x >> 1) & one, b3 = (x >> 3) & one;
  const uint64_t cg = b1 & ~b3, at = ~b1 & one;
  return x ^ (cg << 2) ^ (at | (at << 2) | (at << 4));
}
// dst[0 .. len) = upc(src[0 .. len)), quadwords + a bytewise tail (no read or write beyond len); returns nonzero when a
// letter outside A, C, G, T, N was copied (RAW = no upper-casing: the consensus is compared as it is)
template <bool RAW>
__device__ __forceinline__ int copy_letters8(uint8_t* dst, const uint8_t* src, int len, int lane) {
  uint64_t bad = 0;
  const int full = len & ~7;
  for (int i = lane * 8; i < full; i += WAVE * 8) {
    uint64_t v = ld8u(src + i);
    if (!RAW) v = upc8(v);
    bad |= not_acgtn8(v);
    st8u(dst + i, v);
  }
  if (lane < len - full) {
    uint8_t c = src[full + lane];
    if (!RAW) c = upc(c);
    bad |= comp_acgtn(c) ? 0ull : 1ull;
    dst[full + lane] = c;
  }
  return bad != 0ull;
}
// dst = reverse complement of src[0 .. len) (letters in A, C, G, T, N), LDS to LDS
__device__ __forceinline__ void revcomp8(uint8_t* dst, const uint8_t* src, int len, int lane) {
  const int full = len & ~7;
  for (int i = lane * 8; i < full; i += WAVE * 8) st8u(dst + i, comp8(__builtin_bswap64(sp_lds8a(src + len - 8 - i))));   // (LDS: aligned dwords only, see sp_lds8a)
  if (lane < len - full) dst[full + lane] = comp_acgtn(src[len - 1 - full - lane]);
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

__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
  return ((uint64_t)(uint32_t)rfl((int)(v >> 32)) << 32) | (uint32_t)rfl((int)(v & 0xffffffffull));
}

// per-junction uniform state carried between the stages.  Every field is passed through
// v_readfirstlane (uniformize()) so that hipcc sees the stage-level control flow as
// wave-uniform and emits scalar branches: its exec-mask structurization of this (uniform, but
// not provably so) control flow produced a non-terminating loop on gfx950.
struct JCtx {
  int j;
  int m, n;
  int svt, svS, svE;
  int sBeg, sEnd, eBeg, eEnd;
  bool go, direct;
  int dirty;          // FAST setup only: a letter outside A, C, G, T, N (or lower case in the consensus) was seen
  uint8_t* ob;
  uint64_t ob_off;
  dellyhip_result* out;
  int consLeft, refLeft, refRight, consRight;
  __device__ __forceinline__ void uniformize() {
    j = rfl(j); m = rfl(m); n = rfl(n); svt = rfl(svt); svS = rfl(svS); svE = rfl(svE);
    sBeg = rfl(sBeg); sEnd = rfl(sEnd); eBeg = rfl(eBeg); eEnd = rfl(eEnd);
    go = rfl((int)go) != 0; direct = rfl((int)direct) != 0;
    dirty = rfl(dirty);
    ob = reinterpret_cast<uint8_t*>(rfl64(reinterpret_cast<uint64_t>(ob)));
    ob_off = rfl64(ob_off);
    out = reinterpret_cast<dellyhip_result*>(rfl64(reinterpret_cast<uint64_t>(out)));
    consLeft = rfl(consLeft); refLeft = rfl(refLeft); refRight = rfl(refRight); consRight = rfl(consRight);
  }
};

// _initBreakpoint (tags.h:151-172) + the pieces _getSVRef (split.h:70-163) concatenates.
// Returns false for an unknown svt (_getSVRef returns "").
template <bool INS>
__device__ __forceinline__ bool window_segments(const SplitArgs& A, const dellyhip_junction& J, int m, Seg (&seg)[3],
                                                int& nseg, int& sBeg, int& sEnd, int& eBeg, int& eEnd) {
  const dellyhip_params& P = A.p;
  bool go = true;
  nseg = 0;
  const int boundary = m;
  const int svS = J.sv_start, svE = J.sv_end;
  const int len1 = (int)(uint32_t)A.chr_len[J.chr], len2 = (int)(uint32_t)A.chr_len[J.chr2];
  const uint8_t* c1 = A.chr_seq[J.chr];
  const uint8_t* c2 = A.chr_seq[J.chr2];
  if (INS) {
    // split.h:650-652: bufferSpace in size_t arithmetic, then (int32_t); tags.h:153-157; split.h:122
    const int bs = max((int)(int32_t)(((uint64_t)(int64_t)m - (uint64_t)(int64_t)J.ins_len) / 3ull), P.minimum_flank_size);
    sBeg = max(0, svS - bs);
    sEnd = min(len1, svS + bs);
    eBeg = max(0, svE - bs);
    eEnd = min(len2, svE + bs);
    seg[0] = Seg{c1, sBeg, max(0, eEnd - sBeg), 0};
    nseg = 1;
  } else if (is_tra(J.svt)) {
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
      go = false;  // unknown svt: _getSVRef returns ""
    }
  }
  return go;
}

// ---- stage 1 ---------------------------------------------------------------------
// INS: the caller is the insertion kernel (svt 4, splitAlign path); every other kernel flags
// svt 4 junctions, which the host never routes to them, with DELLYHIP_E_LIMIT.
// FAST (split_sparse_kernel): quadword copies; the reverse complements assume clean letters, X.dirty tells the caller
// when they are not (it then leaves the junction to a kernel with the exact byte-wise semantics).
template <int K, bool WRITE_DEFAULTS = true, typename STR = StrLds, bool INS = false, bool FAST = false>
__device__ __forceinline__ void junction_setup(const SplitArgs& A, int j, STR& S, JCtx& X, int lane) {
  const dellyhip_junction J = A.junc[j];
  const dellyhip_params& P = A.p;
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
  X.direct = (A.ref_base != nullptr);
  X.dirty = 0;
  X.consLeft = X.refLeft = X.refRight = X.consRight = 0;
  const int m = X.m;
  const uint8_t* cons_g = A.cons_base + A.cons_off[j];
  // WRITE_DEFAULTS: first visit of this junction (status/sr_support come from the MSA stage);
  // otherwise a later kernel re-derives the strings and must not touch the result record
  const int prior = WRITE_DEFAULTS ? X.out->status : 0;
  const int support = WRITE_DEFAULTS ? X.out->sr_support : 0;
  int status = 0;
  bool go = true;
  bool mlimit = false;
  if (prior) {
    status = prior;
    mlimit = true;
    go = false;
  } else if (m < 0 || m > MMAX || m + 1 > WAVE * K || m + 1 > STR::cons_cap) {
    status = DELLYHIP_E_LIMIT;
    mlimit = true;
    go = false;
  }
  int dirty = 0;
  // FAST: the consensus is fetched further down, together with the window (every load of the junction's letters in flight at
  // once: copy loops that load, wait and store cost one memory round trip per pass -- consensus, its copy into the result
  // slot and two or three passes over the window were five of them, most of this stage's time)
  const bool go_cons = go;
  if (go && FAST) {
    static_assert(!FAST || STR::cons_cap <= WAVE * 8, "one quadword of the consensus per lane");
  } else if (go) {
    for (int i = lane; i < m; i += WAVE) {
      uint8_t ch = cons_g[i];
      S.cons[i] = ch;
      if (WRITE_DEFAULTS && A.cons_base != A.out_blob) X.ob[i] = ch;   // (MSA modes: the consensus already lives in the slot)
    }
  }
  if (go && !X.direct && (J.svt == 4) != INS) {  // splitAlign/edlib path <-> insertion kernel only
    status = DELLYHIP_E_LIMIT;
    go = false;
  }
  if (go && !X.direct && !(P.reserved & 2) && m < 2 * P.minimum_flank_size + J.ins_len) go = false;  // split.h:647 (bit 1 of reserved: _generateProbes has no such test)

  // _initBreakpoint (tags.h:151-172) + _getSVRef segments (split.h:70-163)
  Seg seg[3];
  int nseg = 0, n = 0;
  if (go && X.direct) {
    n = A.ref_len[j];
    if (n > STR::ref_cap || n < 0) { status = DELLYHIP_E_LIMIT; go = false; }
    else {
      const uint8_t* rg = A.ref_base + A.ref_off[j];
      for (int i = lane; i < n; i += WAVE) S.ref[i] = rg[i];
    }
  } else if (go) {
    int sBeg, sEnd, eBeg, eEnd;
    if (!window_segments<INS>(A, J, m, seg, nseg, sBeg, sEnd, eBeg, eEnd)) go = false;
    X.sBeg = sBeg; X.sEnd = sEnd; X.eBeg = eBeg; X.eEnd = eEnd;
    // (the loops over the <= 3 segments are unrolled with constant indices: a dynamically indexed Seg array lives in
    //  scratch memory -- every lane of every wavefront stored and re-loaded it through HBM)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < nseg) n += seg[q].len;
    if (go && (n > STR::ref_cap || (INS && n < 3))) {  // (splitAlign indexes distRev[n-2]: n < 3 is outside its domain)
      status = DELLYHIP_E_LIMIT;
      go = false;
    }
    if (go && FAST) {
      // quadword t of segment q: letters [8 lane + 512 t, + 8); the tail of a segment (len % 8 letters) one byte per lane
      constexpr int T = (STR::ref_cap + WAVE * 8 - 1) / (WAVE * 8);
      uint64_t wv[3][T], cv = 0;
      uint8_t wt[3] = {0, 0, 0}, ct = 0;
      const int cfull = m & ~7;
      if (lane * 8 < cfull) cv = ld8u(cons_g + lane * 8);
      if (lane < m - cfull) ct = cons_g[cfull + lane];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int t = 0; t < T; ++t) wv[q][t] = 0;
        if (q < nseg && !seg[q].rc) {
          const uint8_t* src = seg[q].base + seg[q].beg;
          const int full = seg[q].len & ~7;
#pragma unroll
          for (int t = 0; t < T; ++t)
            if (lane * 8 + t * WAVE * 8 < full) wv[q][t] = ld8u(src + lane * 8 + t * WAVE * 8);
          if (lane < seg[q].len - full) wt[q] = src[full + lane];
        }
      }
      uint64_t bad = 0;
      if (lane * 8 < cfull) {
        bad |= not_acgtn8(cv);
        st8u(S.cons + lane * 8, cv);
        if (WRITE_DEFAULTS && A.cons_base != A.out_blob) st8u(X.ob + lane * 8, cv);
      }
      if (lane < m - cfull) {
        bad |= comp_acgtn(ct) ? 0ull : 1ull;
        S.cons[cfull + lane] = ct;
        if (WRITE_DEFAULTS && A.cons_base != A.out_blob) X.ob[cfull + lane] = ct;
      }
      int o = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < nseg) {
          const Seg sg = seg[q];
          if (!sg.rc) {
            const int full = sg.len & ~7;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              if (lane * 8 + t * WAVE * 8 < full) {
                const uint64_t v = upc8(wv[q][t]);
                bad |= not_acgtn8(v);
                st8u(S.ref + o + lane * 8 + t * WAVE * 8, v);
              }
            }
            if (lane < sg.len - full) {
              const uint8_t c = upc(wt[q]);
              bad |= comp_acgtn(c) ? 0ull : 1ull;
              S.ref[o + full + lane] = c;
            }
          } else {   // (a reverse-complemented piece keeps letters outside A, C, G, T, N: check what was written)
            fill_segment(S.ref + o, sg, lane);
            DH_SYNC();
            for (int i = lane; i < sg.len; i += WAVE) dirty |= comp_acgtn(S.ref[o + i]) ? 0 : 1;
          }
          o += sg.len;
        }
      }
      dirty |= bad != 0ull;
    } else if (go) {
      int o = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < nseg) {
          const Seg sg = seg[q];
          if (FAST && !sg.rc) dirty |= copy_letters8<false>(S.ref + o, sg.base + sg.beg, sg.len, lane);
          else {
            fill_segment(S.ref + o, sg, lane);
            if (FAST) {    // (a reverse-complemented piece keeps letters outside A, C, G, T, N: check what was written)
              DH_SYNC();
              for (int i = lane; i < sg.len; i += WAVE) dirty |= comp_acgtn(S.ref[o + i]) ? 0 : 1;
            }
          }
          o += sg.len;
        }
      }
    }
  }
  if (FAST && go_cons && !(go && !X.direct)) {   // (no window was fetched: alignConsensus's early exits, limits, direct mode -- the consensus alone)
    dirty |= copy_letters8<true>(S.cons, cons_g, m, lane);
    if (WRITE_DEFAULTS && A.cons_base != A.out_blob) {
      const int full = m & ~7;
      for (int i = lane * 8; i < full; i += WAVE * 8) st8u(X.ob + i, ld8u(cons_g + i));
      if (lane < m - full) X.ob[full + lane] = cons_g[full + lane];
    }
  }
  X.n = n;
  X.go = go;
  X.dirty = FAST ? (__ballot(dirty != 0) != 0ull) : 0;
  // result defaults (everything a later stage does not overwrite): lane q holds dword q of the record and ONE coalesced
  // store writes it (a local dellyhip_result filled through a pointer lived in scratch memory)
  if (WRITE_DEFAULTS) {
    static_assert(sizeof(dellyhip_result) % 4 == 0 && sizeof(dellyhip_result) / 4 <= WAVE, "record layout");
    auto at = [](size_t byte_offset) { return (int)(byte_offset / 4); };
    int v = 0;
    v = (lane == at(offsetof(dellyhip_result, svid))) ? J.svid : v;
    v = (lane == at(offsetof(dellyhip_result, sv_start))) ? J.sv_start : v;
    v = (lane == at(offsetof(dellyhip_result, sv_end))) ? J.sv_end : v;
    v = (lane == at(offsetof(dellyhip_result, ins_len))) ? J.ins_len : v;
    v = (lane == at(offsetof(dellyhip_result, score_unsplit)) || lane == at(offsetof(dellyhip_result, score_best)) ||
         lane == at(offsetof(dellyhip_result, cons_left)) || lane == at(offsetof(dellyhip_result, ref_left)) ||
         lane == at(offsetof(dellyhip_result, ref_right)) || lane == at(offsetof(dellyhip_result, matches)) ||
         lane == at(offsetof(dellyhip_result, mismatches))) ? -1 : v;
    v = (lane == at(offsetof(dellyhip_result, cons_len))) ? (mlimit ? 0 : m) : v;
    v = (lane == at(offsetof(dellyhip_result, cons_off))) ? (int)(uint32_t)(X.ob_off & 0xffffffffull) : v;
    v = (lane == at(offsetof(dellyhip_result, cons_off)) + 1) ? (int)(uint32_t)(X.ob_off >> 32) : v;
    v = (lane == at(offsetof(dellyhip_result, sr_support))) ? support : v;
    v = (lane == at(offsetof(dellyhip_result, status))) ? status : v;
    v = (lane == at(offsetof(dellyhip_result, ref_len))) ? n : v;
    if (lane < (int)(sizeof(dellyhip_result) / 4)) reinterpret_cast<int*>(X.out)[lane] = v;
  }
  DH_SYNC();
  if constexpr (STR::has_rc) {
    if (go && FAST) {
      if (!X.dirty) {
        revcomp8(S.rcons, S.cons, m, lane);
        revcomp8(S.rref, S.ref, n, lane);
      }
    } else if (go) {
      // reverseComplement(s1), reverseComplement(s2): util.h:549-563
      for (int i = lane; i < m; i += WAVE) S.rcons[i] = rc_at(S.cons, m, i);
      for (int i = lane; i < n; i += WAVE) S.rref[i] = rc_at(S.ref, n, i);
    }
    DH_SYNC();
  }
  X.uniformize();
}

// ---- stage 3: checks, join winner, refRight ------------------------------------------
// unsplit = mat[m][n], revmn = rev[m][n]; key = (sum' << 32) | (slot << 12) | (4095 - col).
// code_word(slot, t0) returns the group of 2-bit codes of row `slot` that holds zero-based
// producer step t0 (fields_per_word steps per group, field index = t0 % fields); it is called
// with a per-lane t0.
template <int K, typename CodeWord>
__device__ __forceinline__ void junction_finish(JCtx& X, int unsplit, int revmn, long long key, CodeWord code_word,
                                                int fields_per_word, int lane) {
  if (!X.go) return;
  const int m = X.m, n = X.n;
  // the key comes out of a shuffle reduction: make it provably wave-uniform so that the
  // loop below is a scalar loop (hipcc's divergent-loop lowering of it around
  // v_readfirstlane never terminated on gfx950)
  const int khi = rfl((int)(key >> 32)), klo = rfl((int)(key & 0xffffffffll));
  int best = khi - m;  // sum' = sum + m
  int sstar = (int)(((unsigned)klo >> 12) & 0xfffffu);
  int refLeft = 4095 - (klo & 4095);
  unsplit = rfl(unsplit);
  revmn = rfl(revmn);
  int consRight = sstar, consLeft = m - sstar;
  int refRight = 0;
  bool found = false;
  if (unsplit == revmn) {  // needle.h:83-85
    if (best <= unsplit) {  // no improving join: bestScore stays mat[m][n]
      best = unsplit;
      consLeft = 0;
      refLeft = 0;
      consRight = m;
    }
    // refRight: last t in [0, n-refLeft] with rev[consRight][t] == running max there
    // (needle.h:119-123) = the highest column <= Xc whose code is non-zero (new max or tie).
    // Lane q inspects the q-th code word below the one that holds column Xc; no
    // data-dependent loop exit (a serial scan with early exit was mis-lowered by hipcc:
    // its divergent-loop form never terminated on gfx950).
    {
      const int F = fields_per_word;
      const int ls = consRight / K;
      const int Xc = n - refLeft;
      const int t = Xc + ls - 1;             // zero-based producer step of column Xc
      const int wtop = (t >= 0) ? t / F : -1;
      const int rounds = (Xc >= 1) ? (Xc + F * WAVE - 1) / (F * WAVE) + 1 : 0;
      int bestcol = 0;
      for (int r = 0; r < rounds; ++r) {
        const int widx = wtop - (r * WAVE + lane);
        int cand = 0;
        if (widx >= 0) {
          const uint32_t w = code_word(consRight, widx * F);
          const int fmax = min(F - 1, t - widx * F);   // column <= Xc
          const int fmin = max(0, ls - widx * F);      // column >= 1
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
    found = (best != unsplit);  // needle.h:152
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

// direction pass + traceback with the smallest rows-per-lane that covers rows 0..rmax
template <int K>
__device__ __forceinline__ int dir_and_trace(const uint8_t* rowstr, const uint8_t* colstr, int m, int rmax, int ncols,
                                             uint32_t* scratch, uint8_t* tr, int lane, int& tailV, int& tailH) {
  const int kd = (rmax + 1 + WAVE - 1) / WAVE;  // <= K
  int n = 0;
#define DH_DIR_CASE(KD)                                              \
  if (KD <= K && kd == KD) {                                         \
    pass_dir<KD>(rowstr, colstr, m, rmax, ncols, scratch, lane);     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
    n = traceback<KD>(scratch, rmax, ncols, tr, lane, tailV, tailH); \
  }
  DH_DIR_CASE(1) DH_DIR_CASE(2) DH_DIR_CASE(3) DH_DIR_CASE(4) DH_DIR_CASE(5)
#undef DH_DIR_CASE
  return n;
}

// column masks -> cumulative counts, equality mask, optional alignment rows.  Columns
// [0, posC) take their letters from S.cons / S.ref by running count, columns [posC, Ltot) from
// the reverse-complemented strings through needle.h:209-217's output switch.
template <typename STRS, typename PL>
__device__ __forceinline__ void masks_finish(const SplitArgs& A, JCtx& X, STRS& S, PL& L, int Ltot, int posC,
                                             int lane) {
  const int m = X.m, n = X.n;
  uint8_t* ob = X.ob;
  DH_SYNC();
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
  DH_SYNC();
  // characters of every column -> equality mask (+ optional alignment output)
  uint8_t* aln = ob + A.out_cons_cap + A.out_allele_cap;
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
        if (v) c0 = S.cons[cv];
        if (r) c1 = S.ref[cr];
      } else {
        if (v) c0 = outmap(S.rcons[m - 1 - cv]);
        if (r) c1 = outmap(S.rref[n - 1 - cr]);
      }
    }
    unsigned long long e = __ballot(jcol < Ltot && v && r && c0 == c1);
    if (lane == 0) L.mE[w] = e;
    if (A.want_alignment && jcol < Ltot) {
      aln[jcol] = c0;
      aln[Ltot + jcol] = c1;
    }
  }
  DH_SYNC();
  if (A.want_alignment && lane == 0) {
    X.out->aln_off = X.ob_off + A.out_cons_cap + A.out_allele_cap;
    X.out->aln_len = Ltot;
  }
}

// _findSplit / _percentIdentity / _findHomology / _coordTransform / exact alleles on the column
// masks (split.h:166-375, 596-637); writes the result record.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// MID: called exactly once on every path, behind the homology stage when there is an alignment to look at (split_sparse_kernel
// parks the answer of its work-counter atomic there: by then the round trip is over, and the register that waits for it is free
// for the allele copies)
template <typename STRS, typename PL, typename MID = NoHook>
// pre_ma / pre_mm >= 0: the match / mismatch column counts are known (L.mE is not read).
// clean_letters: consensus and window hold A, C, G, T, N only -- every alignment column then shows the letters of
// S.cons / S.ref in their order (the output switch of needle.h:209-217 undoes the reverse complement), so the two
// alleles are plain substrings and are copied eight letters at a time instead of column by column.
// MR.valid: the masks in registers (sparse_masks_regs: nothing of L is read then); without it they are fetched from L once when the
// alignment has fewer than 64 x 64 columns (MaskRegs, split_kernel.hpp), longer alignments walk the LDS masks.
__device__ __forceinline__ void split_detect(const SplitArgs& A, JCtx& X, STRS& S, PL& L, bool go, int Ltot,
                                             int posC, int lane, int pre_ma = -1, int pre_mm = -1, bool clean_letters = false,
                                             const MaskRegs MR = MaskRegs{0ull, 0ull, 0, 0, false}, MID mid = MID()) {
  const dellyhip_params& P = A.p;
  const int m = X.m, n = X.n;
  uint8_t* ob = X.ob;
  // _findSplit (split.h:319-375) on the masks (uniform code)
#ifdef DH_SPS_FINE
  unsigned long long td[6];
  td[0] = td[1] = td[2] = td[3] = td[4] = td[5] = wall_clock64();
#endif
  if (go) {
    const int svt = X.svt;
    const bool regs = MR.valid || Ltot <= MASKREG_COLS;
    MaskRegs M = MR;   // (by value: a MaskRegs whose address is taken lives in scratch memory)
    if (!MR.valid && regs) {
      const int nw = (Ltot + 63) >> 6;
      M.mv = (lane < nw) ? L.mV[lane] : 0ull;
      M.mr = (lane < nw) ? L.mR[lane] : 0ull;
      M.cv = L.cumV[min(lane, nw)];
      M.cr = L.cumR[min(lane, nw)];
    }
    int cStart = 0, cEnd = 0, rStart = 0, rEnd = 0;
    int closedLen = 0, chosenLen = 0;
    if (regs) {
      const unsigned long long inside = (lane < (Ltot >> 6)) ? ~0ull : (lane == (Ltot >> 6)) ? ((1ull << (Ltot & 63)) - 1ull) : 0ull;
      const unsigned long long bothw = M.mv & M.mr & inside, gapw = ~(M.mv & M.mr) & inside;
      const int fv = next_set_reg(M.mv & inside, 0, Ltot, lane), fr = next_set_reg(M.mr & inside, 0, Ltot, lane);
      int pos = max(fv, fr);  // first column with varIndex > 0 && refIndex > 0
      while (pos < Ltot) {
        const int a = next_set_reg(gapw, pos, Ltot, lane);    // next gap column: NOT (v & r)
        if (a >= Ltot) break;
        const int b1 = next_set_reg(bothw, a, Ltot, lane);    // end of the run: next non-gap column
        if (b1 >= Ltot) break;  // trailing run: never closed, never evaluated
        const int ra = cnt_before_reg(M.mr, M.cr, a), rb = cnt_before_reg(M.mr, M.cr, b1);
        const int va = cnt_before_reg(M.mv, M.cv, a), vb = cnt_before_reg(M.mv, M.cv, b1);
        const int refspan = rb - ra + 1, varspan = vb - va + 1;
        closedLen += b1 - a;
        const bool better = (svt == 4) ? (varspan > (cEnd - cStart)) : (refspan > (rEnd - rStart));
        if (better) {
          rStart = ra; rEnd = ra + refspan; cStart = va; cEnd = va + varspan;
          chosenLen = b1 - a;
        }
        pos = b1 + 1;
      }
    } else {
    int fv = next_set(L.mV, 0ull, 0, Ltot), fr = next_set(L.mR, 0ull, 0, Ltot);
    int J0 = max(fv, fr);  // first column with varIndex > 0 && refIndex > 0
    int pos = J0;
    while (pos < Ltot) {
      int a = pos;  // next gap column: NOT (v & r)
      while (a < Ltot) {
        int w = a >> 6, o = a & 63;
        unsigned long long x = (~(L.mV[w] & L.mR[w])) >> o;
        if (x) { a += __builtin_ctzll(x); break; }
        a = (w + 1) << 6;
      }
      if (a >= Ltot) break;
      int b1 = a;  // end of the run: next non-gap column
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
        chosenLen = b1 - a;
      }
      pos = b1 + 1;
    }
    }
#ifdef DH_SPS_FINE
    td[1] = wall_clock64();
#endif
    bool ok = rEnd > rStart;
    if (ok) {
      if (svt == 4) ok = ((rEnd - rStart) < 5) && ((cEnd - cStart) > 15);
      else ok = ((cEnd - cStart) < 5) && ((rEnd - rStart) > 15);
    }
    int ma = 0, mm = 0;
    float percId = 0.f;
    if (ok) {
      // _percentIdentity split.h:282-316
      if (pre_ma >= 0) {
        ma = pre_ma;
        mm = pre_mm;
      } else {
        for (int w = 0; w < ((Ltot + 63) >> 6); ++w) {
          unsigned long long both = L.mV[w] & L.mR[w];
          ma += __popcll(both & L.mE[w]);
          mm += __popcll(both & ~L.mE[w]);
        }
      }
      mm += closedLen - chosenLen;
      percId = (float)(uint32_t)ma / (float)(uint32_t)(ma + mm);
      if (percId < P.flank_quality) ok = false;
    }
#ifdef DH_SPS_FINE
    td[2] = wall_clock64();
#endif
    int homLeft = 0, homRight = 0;
    if (ok) {
      // _findHomology split.h:262-280
      constexpr bool LDSSTR = std::is_array<typename std::remove_reference<decltype(S.cons)>::type>::value;   // (StrPtr: strings in the workspace)
      if (svt == 4) {
        homRight = longest_homology<LDSSTR>(S.cons, cStart, 1, m - cStart, S.ref, rEnd - 1, 1, n - (rEnd - 1));
        homLeft = longest_homology<LDSSTR>(S.cons, cEnd - 2, -1, min(cEnd - 1, m), S.ref, rStart - 1, -1, min(rStart, n));
      } else {
        homRight = longest_homology<LDSSTR>(S.cons, cEnd - 1, 1, m - (cEnd - 1), S.ref, rStart, 1, n - rStart);
        homLeft = longest_homology<LDSSTR>(S.cons, cStart - 1, -1, min(cStart, m), S.ref, rEnd - 2, -1, min(rEnd - 1, n));
      }
      const int varIndex = m, refIndex = n;
      if ((homLeft + P.minimum_flank_size > cStart) || (varIndex < cEnd + homRight + P.minimum_flank_size)) ok = false;
      if ((homLeft + P.minimum_flank_size > rStart) || (refIndex < rEnd + homRight + P.minimum_flank_size)) ok = false;
    }
#ifdef DH_SPS_FINE
    td[3] = wall_clock64();
#endif
    mid();
    if (ok) {
      // _coordTransform split.h:166-244
      uint32_t gs = 0, ge = 0;
      bool ct_ok = true;
      const int svS = X.svS, svE = X.svE;
      const int sBeg = X.sBeg, sEnd = X.sEnd, eBeg = X.eBeg, eEnd = X.eEnd;
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
      } else if (svt == 4) {
        gs = (uint32_t)(sBeg + rStart);
        ge = (uint32_t)(sBeg + rEnd);
      }
#ifdef DH_SPS_FINE
      td[4] = wall_clock64();
#endif
      int allele_len = 0, status = 0;
      const bool final_ok = ct_ok && (is_tra(svt) || gs < ge);
      if (final_ok) {
        // exact alleles split.h:606-624
        if ((svE - svS <= P.indelsize) && (svt == 2 || svt == 4)) {
          // The chosen run [a, b1) lies between two columns that carry both letters (a - 1 and b1), so the cStart-th consensus letter
          // sits in column a - 1 and the cEnd-th in column b1: REF = ref[rStart - 1 .. rEnd - 1), ALT = cons[cStart - 1 .. cEnd - 1)
          // without looking at the masks -- unless the run starts at the first column that has both indices > 0 (cStart or rStart 0).
          int colA = 0, colB = 0, rA, vA, nr, na;
#ifdef DH_NO_DIRECT_CUT   // (verification builds: every junction locates its alleles on the masks, as the rare cStart / rStart == 0 ones do)
          const bool direct_cut = false;
#else
          const bool direct_cut = cStart >= 1 && rStart >= 1;
#endif
          if (direct_cut) {
            rA = rStart - 1; vA = cStart - 1; nr = rEnd - rStart; na = cEnd - cStart;
          } else if (regs) {
            colA = (cStart >= 1) ? select_bit_reg(M.mv, M.cv, cStart, Ltot, lane) : Ltot;
            colB = select_bit_reg(M.mv, M.cv, cEnd, Ltot, lane);
            if (colA > colB) colA = colB;
            rA = cnt_before_reg(M.mr, M.cr, colA); vA = cnt_before_reg(M.mv, M.cv, colA);
            nr = cnt_before_reg(M.mr, M.cr, colB) - rA; na = cnt_before_reg(M.mv, M.cv, colB) - vA;
          } else {
            colA = (cStart >= 1) ? select_bit(L.mV, L.cumV, cStart, Ltot) : Ltot;
            colB = select_bit(L.mV, L.cumV, cEnd, Ltot);
            if (colA > colB) colA = colB;
            rA = cnt_before(L.mR, L.cumR, colA); vA = cnt_before(L.mV, L.cumV, colA);
            nr = cnt_before(L.mR, L.cumR, colB) - rA; na = cnt_before(L.mV, L.cumV, colB) - vA;
          }
          uint8_t* al = ob + A.out_cons_cap;
          if ((P.reserved & 4) && clean_letters && nr + na + 1 <= A.out_allele_cap) {
            // compact payload (dellyhip_params.reserved bit 2): the two alleles are plain substrings -- REF = window[rStart-1 .. rEnd-1),
            // ALT = consensus[cStart-1 .. cEnd-1) -- which the caller re-cuts from its own copy of the chromosome and the consensus
            // (dellyhip_recut_alleles); the record carries the NEGATED length, no bytes are written or returned
            allele_len = -(nr + na + 1);
          } else if (nr + na + 1 <= A.out_allele_cap && clean_letters) {
            // REF = S.ref[rA .. rB), ALT = S.cons[vA .. vB)
            const int fr = nr & ~7, fa = na & ~7;
            for (int i = lane * 8; i < fr; i += WAVE * 8) st8u(al + i, sp_lds8a(S.ref + rA + i));
            if (lane < nr - fr) al[fr + lane] = S.ref[rA + fr + lane];
            for (int i = lane * 8; i < fa; i += WAVE * 8) st8u(al + nr + 1 + i, sp_lds8a(S.cons + vA + i));
            if (lane < na - fa) al[nr + 1 + fa + lane] = S.cons[vA + fa + lane];
            if (lane == 0) al[nr] = ',';
            allele_len = nr + na + 1;
          } else if (nr + na + 1 <= A.out_allele_cap) {
            if (direct_cut || regs) {   // (letters outside A, C, G, T, N: column by column from the LDS masks, which every caller with such letters has)
              colA = (cStart >= 1) ? select_bit(L.mV, L.cumV, cStart, Ltot) : Ltot;
              colB = select_bit(L.mV, L.cumV, cEnd, Ltot);
              if (colA > colB) colA = colB;
            }
            for (int base = colA & ~63; base < colB; base += 64) {
              int jcol = base + lane;
              int w = base >> 6;
              unsigned long long mv = L.mV[w], mr = L.mR[w];
              unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
              int cv = L.cumV[w] + __popcll(mv & below);
              int cr = L.cumR[w] + __popcll(mr & below);
              bool v = (mv >> lane) & 1ull, r = (mr >> lane) & 1ull;
              if (jcol >= colA && jcol < colB) {
                if (v) al[nr + 1 + (cv - vA)] = (jcol < posC) ? S.cons[cv] : outmap(S.rcons[m - 1 - cv]);
                if (r) al[cr - rA] = (jcol < posC) ? S.ref[cr] : outmap(S.rref[n - 1 - cr]);
              }
            }
            if (lane == 0) al[nr] = ',';
            allele_len = nr + na + 1;
          } else {
            status = DELLYHIP_E_LIMIT;
          }
        }
      }
      if (lane == 0) {
        dellyhip_result* R = X.out;
        R->c_start = cStart; R->c_end = cEnd; R->r_start = rStart; R->r_end = rEnd;
        R->hom_left = homLeft; R->hom_right = homRight;
        R->matches = ma; R->mismatches = mm;
        if (final_ok) {
          if (allele_len) {
            R->allele_off = (allele_len > 0) ? X.ob_off + A.out_cons_cap : 0;
            R->allele_len = allele_len;
          }
          if (status) R->status = status;
          R->ok = 1;
          R->sv_start = (int32_t)gs;
          R->sv_end = (int32_t)ge;
          R->sr_align_quality = percId;
          R->ins_len = cEnd - cStart - 1;
          R->cons_bp = cStart;
          R->hom_len = max(0, homLeft + homRight - 2);
          R->ci_wiggle = max(homLeft, homRight);
        }
#ifdef DH_SPS_FINE
        td[5] = wall_clock64();
        R->score_unsplit = (int)((td[1] - td[0]) / 10);   // findSplit
        R->score_best = (int)((td[2] - td[1]) / 10);      // percent identity
        R->cons_left = (int)((td[3] - td[2]) / 10);       // homology
        R->ref_left = (int)((td[4] - td[3]) / 10);        // coordinates
        R->ref_right = (int)((td[5] - td[4]) / 10);       // alleles + record
#endif
      }
    }
  } else {
    mid();
  }
  DH_SYNC();
}

// longNeedle's glued alignment (needle.h:196-219) as column masks; column order:
// [fwd tail][fwd ops reversed][ref gap][rev ops][rev tail].  trF / trR hold the traceback ops
// in push order (0 's', 1 'v', 2 'h'); tv* / th* are the straight runs at the borders.
// Returns the number of columns, posC = first column of the reverse part.
template <typename PL>
__device__ __forceinline__ int needle_masks(PL& L, const uint8_t* trF, int nF, int tvF, int thF, const uint8_t* trR,
                                            int nR, int tvR, int thR, int gapref, int maskw, int lane, int& posC) {
  for (int w = lane; w < maskw; w += WAVE) {
    L.mV[w] = 0;
    L.mR[w] = 0;
    L.mE[w] = 0;
  }
  DH_SYNC();
  int pos = 0;
  for (int k = 0; k < thF; k += 64) { mask_append(L, pos, min(64, thF - k), 0ull, ~0ull, lane); pos += min(64, thF - k); }
  for (int k = 0; k < tvF; k += 64) { mask_append(L, pos, min(64, tvF - k), ~0ull, 0ull, lane); pos += min(64, tvF - k); }
  for (int k = 0; k < nF; k += 64) {
    int idx = k + lane;
    int op = (idx < nF) ? (int)trF[nF - 1 - idx] : 0;
    unsigned long long v = __ballot(idx < nF && op != 2);
    unsigned long long r = __ballot(idx < nF && op != 1);
    mask_append(L, pos, min(64, nF - k), v, r, lane);
    pos += min(64, nF - k);
  }
  for (int k = 0; k < gapref; k += 64) { mask_append(L, pos, min(64, gapref - k), 0ull, ~0ull, lane); pos += min(64, gapref - k); }
  posC = pos;
  for (int k = 0; k < nR; k += 64) {
    int idx = k + lane;
    int op = (idx < nR) ? (int)trR[idx] : 0;
    unsigned long long v = __ballot(idx < nR && op != 2);
    unsigned long long r = __ballot(idx < nR && op != 1);
    mask_append(L, pos, min(64, nR - k), v, r, lane);
    pos += min(64, nR - k);
  }
  for (int k = 0; k < tvR; k += 64) { mask_append(L, pos, min(64, tvR - k), ~0ull, 0ull, lane); pos += min(64, tvR - k); }
  for (int k = 0; k < thR; k += 64) { mask_append(L, pos, min(64, thR - k), 0ull, ~0ull, lane); pos += min(64, thR - k); }
  return pos;
}

// ---- stage 4: tracebacks + split detection -> result ----------------------------------
template <int K>
__device__ __noinline__ void junction_post(const SplitArgs& A, JCtx& X, StrLds& S, PostLds& L, uint32_t* scratch, int lane) {
  const dellyhip_params& P = A.p;
  const int m = X.m, n = X.n;
  const int consLeft = X.consLeft, refLeft = X.refLeft, consRight = X.consRight, refRight = X.refRight;
  uint8_t* ob = X.ob;
  bool go = X.go;
  // tracebacks (needle.h:154-194) on recomputed direction codes
  int nF = 0, tvF = 0, thF = 0, nR = 0, tvR = 0, thR = 0;
  if (go) {
    if (consLeft > 0 && refLeft > 0) {
      nF = dir_and_trace<K>(S.cons, S.ref, m, consLeft, refLeft, scratch, L.trF, lane, tvF, thF);
    } else {
      tvF = consLeft;
      thF = (consLeft > 0) ? 0 : refLeft;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (consRight > 0 && refRight > 0) {
      nR = dir_and_trace<K>(S.rcons, S.rref, m, consRight, refRight, scratch, L.trR, lane, tvR, thR);
    } else {
      tvR = consRight;
      thR = (consRight > 0) ? 0 : refRight;
    }
  }
  DH_SYNC();

  // alignment as column masks
  int Ltot = 0, posC = 0;
  if (go) {
    Ltot = needle_masks(L, L.trF, nF, tvF, thF, L.trR, nR, tvR, thR, (n - refRight) - refLeft, MASKW, lane, posC);
    masks_finish(A, X, S, L, Ltot, posC, lane);
  }
  if (go && X.direct && lane == 0) X.out->ok = 1;  // longNeedle() returned true
  split_detect(A, X, S, L, go && !X.direct, Ltot, posC, lane);
}

// ---- one junction per wavefront -----------------------------------------------------
struct __attribute__((aligned(16))) WaveLds {
  StrLds s;
  PostLds p;
};

template <int K>
__device__ void process_junction(const SplitArgs& A, int j, WaveLds& L, uint32_t* scratch, int lane) {
  JCtx X;
  junction_setup<K>(A, j, L.s, X, lane);
  int unsplit = 0, revmn = 0;
  long long key = 0;
  if (X.go) {
    const int m = X.m, n = X.n;
    int hfin[K], brfin[K], bestkey[K];
    int hrow_m;
    pass_R<K>(L.s, m, n, scratch, lane, hfin, brfin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pass_M<K>(L.s, m, n, scratch, lane, brfin, bestkey, hrow_m);
#pragma unroll
    for (int i = 0; i < K; ++i)
      if (lane * K + i == m) revmn = hfin[i];
    revmn = __shfl(revmn, m / K) - m;
    unsplit = (__shfl(hrow_m, 0) >> SCALE_SHIFT) - m;
    // first (row, col) in row-major order attaining the maximum (needle.h:107-115):
    // max sum', then max slot (= min row), then max cinv (= min col)
    key = (long long)0x8000000000000000ll;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      int s = lane * K + i;
      if (s <= m) {
        long long kk = ((long long)(bestkey[i] >> SCALE_SHIFT) << 32) | ((long long)s << 12) | (bestkey[i] & 4095);
        key = kk > key ? kk : key;
      }
    }
    key = wave_max64(key);
  }
  auto code_word = [&](int slot, int t) -> uint32_t {
    int ls = slot / K, is = slot - ls * K;
    return ld_scratch(&scratch[((size_t)(t >> 4) * K + is) * WAVE + ls]);
  };
  junction_finish<K>(X, unsplit, revmn, key, code_word, 16, lane);
  junction_post<K>(A, X, L.s, L.p, scratch, lane);
}

template <int K>
__global__ __launch_bounds__(WAVE) void split_align_kernel(SplitArgs A0) {
  __shared__ WaveLds L;
  const int lane = threadIdx.x;
  if (A0.pair_mode && *A0.work_counter == 0) return;  // nothing was deferred by the packed kernel
  if (A0.sps_left && *A0.sps_left == 0) return;
  // (junction_post is a called function and takes the arguments by reference: the copy it reads lives in scratch memory, 208
  //  bytes per lane.  Made HERE, behind the early exits: as the kernel parameter itself it was written at kernel entry -- 13 KB
  //  of HBM writes per wavefront, 78 MB per step, by kernels that return at once because the sparse kernel left nothing)
  const SplitArgs A = A0;
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    const int j = A.work_list[w];
    if (j < 0) continue;
    if (A.pair_mode) {  // launched behind the packed kernel: only junctions it deferred
      if (A.res[j].status != DH_DEFERRED) continue;
      DH_SYNC();
      if (lane == 0) A.res[j].status = 0;
      DH_SYNC();
    }
    process_junction<K>(A, j, L, scratch, lane);
    if (A.pair_mode && lane == 0) A.res[j].reserved = 1;  // post-processing already done
  }
}

// post-processing kernel: one junction per wavefront, junctions the DP kernel marked JS_FOUND
template <int K>
__global__ __launch_bounds__(WAVE) void split_post_kernel(SplitArgs A0) {
  __shared__ WaveLds L;
  const int lane = threadIdx.x;
  if (A0.sps_left && *A0.sps_left == 0) return;
  const SplitArgs A = A0;   // (the copy junction_post reads: made behind the early exit, see split_align_kernel)
  uint32_t* scratch = A.scratch + (size_t)blockIdx.x * A.scratch_words;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    const int j = A.work_list[w];
    if (j < 0) continue;
    const dellyhip_result* r = &A.res[j];
    const int st = r->status, rsv = r->reserved, sb = r->score_best, su = r->score_unsplit;
    if (rsv) {  // finished by the 32-bit kernel
      if (lane == 0) A.res[j].reserved = 0;
      continue;
    }
    if (st != 0 || sb == -1 || sb == su) continue;  // longNeedle found no split (needle.h:83,152)
    const int cl = r->cons_left, rl = r->ref_left, rr = r->ref_right;
    JCtx X;
    junction_setup<K, false>(A, j, L.s, X, lane);
    X.consLeft = cl;
    X.refLeft = rl;
    X.refRight = rr;
    X.consRight = X.m - cl;
    junction_post<K>(A, X, L.s, L.p, scratch, lane);
  }
}

}  // namespace dh
