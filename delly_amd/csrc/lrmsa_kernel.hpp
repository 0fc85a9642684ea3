// lrmsa_kernel.hpp -- gfx950 device code for msaEdlib (src/assemble.h:383-473), the long-read
// consensus of `delly lr` (non-insertion junctions):
//   all-pairs NW distances              -> myers_pairs_kernel (myers_kernel.hpp)
//   medoid, order, drop the worst 20 %  -> lane-parallel ranks (N <= 16)
//   progressive alignment of each read to the running 2-allele consensus:
//     consensusEdlib (:198-259), edlibAlign(read, consensus, NW, PATH, 20 extended-IUPAC
//     equalities) (:425-444), convertAlignment (:24-88)
//   consensus (src/msa.h:111-173) + trim (:465-469)
//
// One junction per wavefront.  edlib's NW PATH is reproduced exactly: unit-cost DP with the
// equality relation, rows = consensus letters in strips of 320 (lr_kernel.hpp geometry), columns
// = read letters, per-cell op codes with edlib's traceback preference.  For alignments above
// 1 MiB of edlib "alignment data" edlib switches to Hirschberg (src/edlib.cpp:1188-1389): the
// target is split in the middle and the FIRST query index whose forward + backward scores reach
// the optimum is taken; the same split is made here (the two score rows are strip boundary rows),
// recursively, so the op string equals edlib's byte for byte.
#pragma once
#include "ins_kernel.hpp"
#include "lr_kernel.hpp"
#include "msa_kernel.hpp"

namespace dh {

// DH_LR_TIMING builds (tools/lrc_phases.py): where a junction's wavefront spends its time, summed over the launch (100 MHz ticks).
// Slots: 1 seeding (k-mer tables, diagonal votes), 2 superstring NW paths (whole), 3 buildSuperstring, 4 column votes (consensusWfa /
// consensusEdlib), 5 forward location pass, 6 reverse location pass, 7 Hirschberg last-row passes, 8 direction fill of the base
// rectangles, 9 tracebacks + op reversal, 10 convertAlignment, 11 final consensus + trimming, 12 Hirschberg split search,
// 13 progressive NW / HW paths (whole), 14 junctions, 15 whole junction
#ifdef DH_LR_TIMING
__device__ unsigned long long dh_lrt[32];
__device__ __forceinline__ unsigned long long& lrt_t0() {
  __shared__ unsigned long long t0;
  return t0;
}
#define LRT_START() do { if (lane == 0) lrt_t0() = wall_clock64(); } while (0)
#define LRT_LAP(slot) do { if (lane == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&dh_lrt[slot], n_ - lrt_t0()); lrt_t0() = n_; } } while (0)
#define LRT_ADD(slot, v) do { if (lane == 0) atomicAdd(&dh_lrt[slot], (unsigned long long)(v)); } while (0)
#else
#define LRT_START() do { } while (0)
#define LRT_LAP(slot) do { } while (0)
#define LRT_ADD(slot, v) do { } while (0)
#endif

constexpr int LM_NR = 32;                 // reads per junction (delly lr: maxReadPerSV = 15 by default, -p; src/tegua.h:241)
constexpr int LM_STACK = 24;              // Hirschberg sub-problems in flight (depth <= log2 of the longer side + 1)
constexpr int LM_RBITS = 15;              // row index bits of the location keys: target rows <= 32766, E < 2^17
constexpr int LM_RMASK = (1 << LM_RBITS) - 1;

struct LrMsaArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  dellyhip_params p;
  dellyhip_result* res;
  uint8_t* out_blob;
  uint64_t out_stride;
  int32_t out_cons_cap;
  int32_t* cons_len;
  const int32_t* edit;      // all-pairs distances, LM_NR x LM_NR per junction
  int32_t n_work;
  uint8_t* ws;              // per resident block
  uint64_t ws_stride;
  int32_t acap;             // alignment columns capacity (<= LR_MMAX + 1)
  int32_t ncap;             // read length capacity
  uint64_t off_alnB, off_astr, off_bnd, off_ops, off_tmp, off_cons, off_dirs;   // alnA at 0
  uint64_t strip_words;
};

struct __attribute__((aligned(16))) LrMsaLds {
  int32_t first[NRMAX], last[NRMAX];      // consensus_node
  int32_t med[LM_NR];
  int32_t sel[LM_NR];
  int32_t rlen[LM_NR];
  uint64_t roff[LM_NR];
};

// index of a letter in the extended-IUPAC equality relation of msaEdlib (src/assemble.h:425), -1 = none:
//   A 0, C 1, G 2, T 3, '-' 4, M 5, R 6, W 7, B 8, S 9, Y 10, D 11, K 12, E 13, F 14
// Round 5: nibble tables in 64-bit constants instead of a `switch` -- hipcc compiles a switch over per-lane letters (like the
// comparison chain of letter_code(), msa_kernel.hpp) into a tree of divergent branches, tens of scalar instructions per letter in
// the mask set-up of every bit-vector pass.
__device__ __forceinline__ int iupac_index(int c) {
  const uint32_t idx = (uint32_t)c - (uint32_t)'A';
  const unsigned long long tbl = (idx & 16u) ? 0xfffffffaf7ff396full : 0xfff5fcfff2edb180ull;
  const int v = (int)((tbl >> (4u * (idx & 15u))) & 15ull);
  const int r = (idx < 32u && v != 15) ? v : -1;
  return (c == '-') ? 4 : r;
}
// bit y of iupac_partners(x): letters x and y are declared equal (symmetric closure of the 20 pairs)
//            A        C        G        T        -
// M={A,C} R={A,G} W={A,T} B={A,-} S={C,G} Y={C,T} D={C,-} K={G,T} E={G,-} F={T,-}
// (A ~ M R W B, C ~ M S Y D, G ~ R S K E, T ~ W Y K F, - ~ B D E F; a pair letter ~ its two members): 16-bit entries, four per constant
__device__ __forceinline__ uint32_t iupac_partners(int x) {
  const uint32_t ux = (uint32_t)x;
  const uint32_t w = ux >> 2;
  const unsigned long long tbl = (w == 0u) ? 0x548032400e2001e0ull : (w == 1u) ? 0x0009000500036900ull : (w == 2u) ? 0x0012000a00060011ull : 0x000000180014000cull;
  const uint32_t v = (uint32_t)((tbl >> (16u * (ux & 3u))) & 0xffffull);
  return (ux < 15u) ? v : 0u;
}

// Unit-cost strip pass with the equality relation.  Target letter of global slot g = q*320 + ls
// is row r = g - pad (rows < 0: +infinity dummies above row 0), t[r-1] = tp[(r-1)*tstep];
// query letters qp[c*qstep].  E[r][0] = r, E[0][c] = c.  DIRS: edlib op code per cell
// (preference INSERT > DELETE > diagonal) into `dirs`; bout: last slot of the strip per column.
struct LmKeys {
  unsigned kf;   // min over rows r0..tlen of (E[r][qlen] << 15) | r            (first optimal end)
  unsigned kl;   // min over rows r0..tlen of (E[r][qlen] << 15) | (32767 - r)  (last optimal end)
};

// mode bits of lm_pass
constexpr int LM_HW = 1;    // E[r][0] = 0 (edlib HW: the query may start anywhere in the target)
constexpr int LM_EQ = 2;    // extended-IUPAC additional equalities

// EQ: 0 plain byte equality; 1 equality classes, every letter of both strings is one of the 15 class
// letters (then "same letter" is "shares a class bit": no byte compare); 2 equality classes + byte compare
// (a string holds a letter outside the 15, e.g. N)
template <bool DIRS, bool LOC, int EQ>
__device__ __noinline__ LmKeys lm_pass_t(const uint8_t* tp, int tstep, int tlen, const uint8_t* qp, int qstep, int qlen, int q,
                                         int pad, int mode, int r0, const int32_t* bin, int32_t* bout, uint32_t* dirs,
                                         int lane, int32_t* lastcol = nullptr) {
  constexpr int K = LRK;
  constexpr int POS = 1 << 28;
  const bool hw = (mode & LM_HW) != 0;
  constexpr bool useeq = EQ != 0;
  int a[K], h[K], colq[K];
  uint32_t part[K], acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int r = q * LRS + lane * K + i - pad;
    a[i] = (r >= 1 && r <= tlen) ? (int)tp[(r - 1) * tstep] : NOMATCH;
    h[i] = (r >= 0) ? (hw ? 0 : r) : POS;
    colq[i] = h[i];
    const int ix = useeq ? iupac_index(a[i]) : -1;
    part[i] = iupac_partners(ix) | ((EQ == 1 && ix >= 0) ? (1u << ix) : 0u);   // (EQ 1: the letter's own bit too)
    acc[i] = 0;
  }
  const int lastrow = min(LRS - 1, tlen + pad - q * LRS);   // local slot of the last real row in this strip
  const int lastlane = lastrow / K;
  const int T = qlen + lastlane;
  const int nblk = (T + 15) >> 4;
  int upPrev = bin ? (hw ? 0 : (q * LRS - pad - 1)) : POS;   // E[row above][0]
  int b = NOMATCH;
  int c = -lane;
  int outv = 0;
  // block blk+1's letters / boundary values are loaded while block blk computes
  // (EQ: the letter travels with its equality-class index in bits 16.., decoded once per 16 steps here
  // instead of once per step in the loop)
  auto ld_chunk = [&](int blk) {
    const int ci = blk * 16 + (lane & 15);
    const int ch = (ci < qlen) ? (int)qp[ci * qstep] : NOMATCH;
    return EQ ? (ch | ((iupac_index(ch) + 1) << 16)) : ch;
  };
  auto ld_bnd = [&](int blk) { const int ci = blk * 16 + (lane & 15); return (bin && ci + 1 <= qlen) ? bin[ci + 1] : POS; };
  int chunk = ld_chunk(0), bchunk = ld_bnd(0);
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_n = ld_chunk(blk + 1), bchunk_n = ld_bnd(blk + 1);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int newc = __builtin_amdgcn_readlane(chunk, f);
      const int bnd = __builtin_amdgcn_readlane(bchunk, f);
      b = dpp_from_prev(b, newc);
      const int recv = dpp_from_prev(h[K - 1], bnd);
      c += 1;
      if ((unsigned)(c - 1) < (unsigned)qlen) {
        const uint32_t ybit = EQ ? ((1u << ((uint32_t)b >> 16)) >> 1) : 0u;   // class index + 1 in bits 16.. (0: none)
        const int bl = EQ ? (b & 0xffff) : b;
        int diag = upPrev, up = recv;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const bool eq = (EQ == 1) ? ((part[i] & ybit) != 0u)
                                    : (EQ == 2) ? ((a[i] == bl) || ((part[i] & ybit) != 0u)) : (a[i] == bl);
          const int x = diag + (eq ? 0 : 1);
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
        if (LOC && c == qlen) {
#pragma unroll
          for (int i = 0; i < K; ++i) colq[i] = h[i];
        }
      }
      upPrev = recv;
      if (bout) outv = writelane16(__builtin_amdgcn_readlane(h[K - 1], 63), f, outv);
    }
    if (DIRS) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        dirs[((size_t)blk * K + i) * WAVE + lane] = acc[i];
        acc[i] = 0;
      }
    }
    if (bout) {
      const int col = blk * 16 + lane - 62;
      if (lane < 16 && col >= 1 && col <= qlen) bout[col] = outv;
    }
    chunk = chunk_n;
    bchunk = bchunk_n;
  }
  LmKeys k;
  k.kf = k.kl = 0xffffffffu;
  if (LOC) {
    unsigned kf = 0xffffffffu, kl = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int r = q * LRS + lane * K + i - pad;
      if (r >= r0 && r <= tlen) {
        kf = min(kf, ((unsigned)colq[i] << LM_RBITS) | (unsigned)r);
        kl = min(kl, ((unsigned)colq[i] << LM_RBITS) | (unsigned)(LM_RMASK - r));
      }
      if (lastcol && r >= 0 && r <= tlen) lastcol[r] = colq[i];   // E[r][qlen] of every row (all optimal end locations: edlib_full_kernel)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      kf = min(kf, (unsigned)__shfl_xor((int)kf, o));
      kl = min(kl, (unsigned)__shfl_xor((int)kl, o));
    }
    k.kf = (unsigned)rfl((int)kf);
    k.kl = (unsigned)rfl((int)kl);
  }
  return k;
}

// the extended-IUPAC equality logic is compiled only into the instances that need it (msaEdlib / msaWfa
// progressive alignments); splitAlign and the superstring use plain byte equality
// true when every letter of s[0..n) (stride `step`) is one of the 15 equality-class letters
__device__ __forceinline__ bool lm_in_classes(const uint8_t* sp, int step, int n, int lane) {
  int bad = 0;
  for (int i = lane; i < n; i += WAVE) bad |= (iupac_index((int)sp[i * step]) < 0) ? 1 : 0;
  return __ballot(bad) == 0ull;
}

// the equality logic is compiled only into the instances that need it (msaEdlib / msaWfa progressive
// alignments); splitAlign and the superstring use plain byte equality.  mode bit LM_EQFAST (set by the
// caller after lm_in_classes) selects the compare-free variant.
constexpr int LM_EQFAST = 4;
template <bool DIRS, bool LOC>
__device__ __forceinline__ LmKeys lm_pass(const uint8_t* tp, int tstep, int tlen, const uint8_t* qp, int qstep, int qlen, int q,
                                          int pad, int mode, int r0, const int32_t* bin, int32_t* bout, uint32_t* dirs,
                                          int lane, int32_t* lastcol = nullptr) {
  if ((mode & LM_EQ) && (mode & LM_EQFAST))
    return lm_pass_t<DIRS, LOC, 1>(tp, tstep, tlen, qp, qstep, qlen, q, pad, mode, r0, bin, bout, dirs, lane, lastcol);
  if (mode & LM_EQ) return lm_pass_t<DIRS, LOC, 2>(tp, tstep, tlen, qp, qstep, qlen, q, pad, mode, r0, bin, bout, dirs, lane, lastcol);
  return lm_pass_t<DIRS, LOC, 0>(tp, tstep, tlen, qp, qstep, qlen, q, pad, mode, r0, bin, bout, dirs, lane, lastcol);
}

// Match masks of one lane-word for the compare-free passes: E[(w * 16 + y) * WAVE + lane] bit q = "row lo + q of the target
// equals query class y" (its own letter or one of the partners).  Eight rows at a time: the eight letters are loaded before any
// of them is used (the row-at-a-time loop of rounds 2-5 waited one HBM / L2 round trip per row: 64-96 in front of EVERY pass),
// and the fifteen masks are built without a branch on the letter.
__device__ __forceinline__ void lm_eq_word(uint32_t* E, int w, gptr_cu8 tp, int tstep, int lo, int tlen, int lane) {
  uint32_t e[15];
#pragma unroll
  for (int y = 0; y < 15; ++y) e[y] = 0;
#pragma unroll 1
  for (int q0 = 0; q0 < 32; q0 += 8) {
    int ch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = lo + q0 + k;
      ch[k] = (r < tlen) ? (int)tp[r * tstep] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = lo + q0 + k;
      const int x = iupac_index(ch[k]);
      const uint32_t pm = (r < tlen && x >= 0) ? ((1u << x) | iupac_partners(x)) : 0u;
#pragma unroll
      for (int y = 0; y < 15; ++y) e[y] |= ((pm >> y) & 1u) << (q0 + k);
    }
  }
#pragma unroll
  for (int y = 0; y < 15; ++y) E[(w * 16 + y) * WAVE + lane] = e[y];
  E[(w * 16 + 15) * WAVE + lane] = 0;
}

// ---- bit-vector flavour of the last-row pass (Hirschberg halves of edlib's NW PATH, edlib.cpp:1163-1389) -------------
// In the compare-free regime (every letter of both strings is one of the 15 classes) "equal" is a relation
// between classes, so the pass is Myers' recurrence with one equality mask per QUERY class: mask[y] = rows whose
// letter x has y in {x} + partners(x).  Layout and column step as myers_nw_fast (myers_kernel.hpp): 32 target rows
// per lane-word, masks in LDS, the query travels through the lanes as mask offsets, no divergent control flow.
// The value of row tlen is read off the owning lane after every column (bottom score of its words minus the
// vertical deltas below row tlen) and leaves through the 16-column staging register of the strip passes.
__device__ __forceinline__ uint32_t* lm_eq_lds() {
  __shared__ uint32_t eqm[MYERS_NW * 16 * WAVE];   // [word][query class 0..14, 15 = none][lane]
  return eqm;
}

template <int NWORDS>
__device__ __noinline__ void lm_last_row_myers(const uint8_t* tp_, int tstep, int tlen, const uint8_t* qp_, int qstep, int qlen,
                                               int32_t* row_out_, int lane) {
  // (the strings and the output row live in HBM: said so, their loads are global_load and may pass the LDS updates of the
  //  match-mask set-up instead of waiting for each of them -- a pointer argument of a called function is generic otherwise)
  const gptr_cu8 tp = (gptr_cu8)tp_, qp = (gptr_cu8)qp_;
  const gptr_i32 row_out = (gptr_i32)row_out_;
  uint32_t* E = lm_eq_lds();
  const int row0 = lane * 32 * NWORDS;
  uint32_t mk[NWORDS];   // rows of this lane at or beyond tlen (their vertical deltas are taken off the bottom score)
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, lo + 32 - tlen));
    mk[w] = (nb >= 32) ? 0xffffffffu : ((nb > 0) ? (~0u << (32 - nb)) : 0u);
    lm_eq_word(E, w, tp, tstep, lo, tlen, lane);
  }
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = 0xffffffffu;
    Mv[w] = 0;
  }
  int score = row0 + 32 * NWORDS;
  const int lastlane = (tlen - 1) / (32 * NWORDS);
  const int T = qlen + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 1;
  int c = -lane;
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    const int y = (ci < qlen) ? iupac_index((int)qp[ci * qstep]) : -1;
    return ((y < 0) ? 15 : y) * WAVE;
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
  int outv = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
      const int newc = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      bs = dpp_from_prev(bs, newc);
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
      int hin = dpp_from_prev(hcarry, 1);
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)qlen;
      uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        uint32_t Eq = EqC[w];
        const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470
        const uint32_t Xv = Eq | Mv[w];
        Eq |= hinNeg;
        const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
        uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
        uint32_t Mh = Pv[w] & Xh;
        const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1u : 0u;
        nP[w] = Mh | ~(Xv | Ph);
        nM[w] = Ph & Xv;
        hin = hout;
      }
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        Pv[w] = valid ? nP[w] : Pv[w];
        Mv[w] = valid ? nM[w] : Mv[w];
      }
      hcarry = valid ? hin : hcarry;
      score += valid ? hin : 0;
      int s = score;   // E[tlen][c] in the lane that owns row tlen
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) s += __popc(Mv[w] & mk[w]) - __popc(Pv[w] & mk[w]);
      const int sv = __builtin_amdgcn_readlane(s, lastlane);
      outv = (lane == f) ? sv : outv;   // (v_writelane_b32 takes one scalar operand only: no scalar lane index next to a scalar value)
    }
    {   // lanes 0..15 hold the columns blk*16 + lane + 1 - lastlane of this block
      const int col = blk * 16 + lane + 1 - lastlane;
      if (lane < 16 && col >= 1 && col <= qlen) row_out[col] = outv;
    }
    chunk = chunk_next;
  }
  if (lane == 0) row_out[0] = tlen;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---- bit-vector flavour of the location passes (edlib HW / SHW: distance + first / last optimal end row) -------------------
// Same layout and column step as lm_last_row_myers.  HW: E[r][0] = 0 (all vertical deltas 0 at column 0), SHW / NW:
// E[r][0] = r (all +1); E[0][c] = c either way.  At the last column every lane is frozen; E[r][qlen] = qlen + the
// prefix sum of the vertical deltas of rows 1..r, which each lane rebuilds for its own rows from Pv / Mv behind a wave
// prefix sum of the lanes' totals.  Keys as lm_pass: (E << LM_RBITS) | r and (E << LM_RBITS) | (LM_RMASK - r) over rows r0..tlen.
template <int NWORDS>
__device__ __noinline__ LmKeys lm_locate_myers(const uint8_t* tp_, int tstep, int tlen, const uint8_t* qp_, int qstep, int qlen, bool hw,
                                               int r0, int lane) {
  const gptr_cu8 tp = (gptr_cu8)tp_, qp = (gptr_cu8)qp_;   // (HBM: see lm_last_row_myers)
  uint32_t* E = lm_eq_lds();
  const int row0 = lane * 32 * NWORDS;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    lm_eq_word(E, w, tp, tstep, lo, tlen, lane);
  }
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = hw ? 0u : 0xffffffffu;
    Mv[w] = 0;
  }
  const int lastlane = (tlen - 1) / (32 * NWORDS);
  const int T = qlen + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 1;
  int c = -lane;
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    const int y = (ci < qlen) ? iupac_index((int)qp[ci * qstep]) : -1;
    return ((y < 0) ? 15 : y) * WAVE;
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
      const int newc = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      bs = dpp_from_prev(bs, newc);
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
      int hin = dpp_from_prev(hcarry, 1);
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)qlen;
      uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        uint32_t Eq = EqC[w];
        const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470
        const uint32_t Xv = Eq | Mv[w];
        Eq |= hinNeg;
        const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
        uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
        uint32_t Mh = Pv[w] & Xh;
        const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1u : 0u;
        nP[w] = Mh | ~(Xv | Ph);
        nM[w] = Ph & Xv;
        hin = hout;
      }
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        Pv[w] = valid ? nP[w] : Pv[w];
        Mv[w] = valid ? nM[w] : Mv[w];
      }
      hcarry = valid ? hin : hcarry;
    }
    chunk = chunk_next;
  }
  // column qlen: vertical deltas of this lane's rows (rows beyond tlen do not count)
  int tot = 0;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, tlen - lo));                       // rows of this word inside the target
    const uint32_t keep = (nb >= 32) ? 0xffffffffu : ((nb > 0) ? ((1u << nb) - 1u) : 0u);
    tot += __popc(Pv[w] & keep) - __popc(Mv[w] & keep);
  }
  int incl = tot;   // inclusive prefix sum over the lanes
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  int val = qlen + incl - tot;   // E[row0][qlen]: the row just above this lane's first row (row 0 for lane 0)
  unsigned kf = 0xffffffffu, kl = 0xffffffffu;
  if (lane == 0 && r0 == 0) {
    kf = ((unsigned)qlen << LM_RBITS) | 0u;
    kl = ((unsigned)qlen << LM_RBITS) | (unsigned)LM_RMASK;
  }
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
#pragma unroll 1
    for (int q = 0; q < 32; ++q) {
      const int r = lo + q + 1;   // 1-based row of bit q
      if (r <= tlen) {
        val += (int)((Pv[w] >> q) & 1u) - (int)((Mv[w] >> q) & 1u);
        if (r >= r0) {
          kf = min(kf, ((unsigned)val << LM_RBITS) | (unsigned)r);
          kl = min(kl, ((unsigned)val << LM_RBITS) | (unsigned)(LM_RMASK - r));
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    kf = min(kf, (unsigned)__shfl_xor((int)kf, o));
    kl = min(kl, (unsigned)__shfl_xor((int)kl, o));
  }
  LmKeys k;
  k.kf = (unsigned)rfl((int)kf);
  k.kl = (unsigned)rfl((int)kl);
  __syncthreads();
  return k;
}

// true when every letter of s[0..n) (stride `step`) is A, C, G, T or '-': among those the extended-IUPAC relation is plain
// identity, so a plain-equality alignment (splitAlign, the superstring) may run the compare-free bit-vector passes
__device__ __forceinline__ bool lm_pure_acgt(const uint8_t* sp, int step, int n, int lane) {
  int bad = 0;
  for (int i = lane; i < n; i += WAVE) {
    const int ix = iupac_index((int)sp[i * step]);
    bad |= (ix < 0 || ix > 4) ? 1 : 0;
  }
  return __ballot(bad) == 0ull;
}

// ---- bit-vector flavour of the traceback-regime path (the base-case rectangles of the Hirschberg split) ----------------
// edlib's direction rule (INSERT > DELETE > diagonal, edlib.cpp:1018-1125) needs per cell only two bits the Myers
// step has in hand: INSERT (left: a query letter alone) <=> the horizontal delta D[r][c] - D[r][c-1] is +1 (Ph),
// DELETE (up: a target letter alone) <=> the vertical delta D[r][c] - D[r-1][c] is +1 (the new Pv); otherwise the move
// is diagonal and MATCH / MISMATCH is the class relation of the two letters.  So the fill stores the two planes
// (2 bits per cell, as the code words do) 32 rows per ~40 instructions, and the windowed run-length traceback reads
// them "transposed": a lane keeps a COLUMN and the 32-row word of its fetch row.
// Layout: word of (step s, word w, owner lane lo) at ((s * NW + w) * (nl + 1) + lo); lanes beyond the last owner
// lane share the dummy slot nl.  Cell (r, c) (1-based): lo = (r-1) / (32 NW), w = ((r-1) >> 5) - lo NW, s = c - 1 + lo.
template <int NWORDS>
__device__ __noinline__ void lm_dirs_myers(const uint8_t* tp_, int tlen, const uint8_t* qp_, int qlen, uint32_t* planeH_,
                                           uint32_t* planeV_, int lane) {
  const gptr_cu8 tp = (gptr_cu8)tp_, qp = (gptr_cu8)qp_;   // (HBM: see lm_last_row_myers)
  const gptr_u32 planeH = (gptr_u32)planeH_, planeV = (gptr_u32)planeV_;
  uint32_t* E = lm_eq_lds();
  const int row0 = lane * 32 * NWORDS;
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    lm_eq_word(E, w, tp, 1, lo, tlen, lane);
  }
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = 0xffffffffu;
    Mv[w] = 0;
  }
  const int lastlane = (tlen - 1) / (32 * NWORDS);
  const int nl1 = lastlane + 2;                       // owner lanes + the shared dummy slot
  const int sl = (lane <= lastlane) ? lane : lastlane + 1;
  const int T = qlen + lastlane;
  const int nblk = (T + 15) >> 4;
  int hcarry = 1;
  int c = -lane;
  auto load_chunk = [&](int blk) -> int {
    const int ci = blk * 16 + (lane & 15);
    const int y = (ci < qlen) ? iupac_index((int)qp[ci]) : -1;
    return ((y < 0) ? 15 : y) * WAVE;
  };
  int chunk = load_chunk(0);
  int bs = dpp_from_prev(0, __builtin_amdgcn_readlane(chunk, 0));
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
  for (int blk = 0; blk < nblk; ++blk) {
    const int chunk_next = load_chunk(blk + 1);
#pragma unroll 1
    for (int f = 0; f < 16; ++f) {
      uint32_t EqC[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
      const int newc = (f == 15) ? __builtin_amdgcn_readlane(chunk_next, 0) : __builtin_amdgcn_readlane(chunk, f + 1);
      bs = dpp_from_prev(bs, newc);
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + bs + lane];
      int hin = dpp_from_prev(hcarry, 1);
      c += 1;
      const bool valid = (unsigned)(c - 1) < (unsigned)qlen;
      const size_t sbase = ((size_t)(blk * 16 + f) * NWORDS) * nl1 + sl;
      uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        uint32_t Eq = EqC[w];
        const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470
        const uint32_t Xv = Eq | Mv[w];
        Eq |= hinNeg;
        const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
        uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
        uint32_t Mh = Pv[w] & Xh;
        planeH[sbase + (size_t)w * nl1] = Ph;            // horizontal +1 deltas of this column's rows
        const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1u : 0u;
        nP[w] = Mh | ~(Xv | Ph);
        nM[w] = Ph & Xv;
        planeV[sbase + (size_t)w * nl1] = nP[w];          // vertical +1 deltas within this column
        hin = hout;
      }
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) {
        Pv[w] = valid ? nP[w] : Pv[w];
        Mv[w] = valid ? nM[w] : Mv[w];
      }
      hcarry = valid ? hin : hcarry;
    }
    chunk = chunk_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// windowed run-length traceback over the two planes (see traceback_runs, split_kernel.hpp, for the row-word
// flavour): lane x keeps column cc - x and the planes' words of the 32-row word its fetch row lies in.
// ops (edlib codes 0 match, 1 insert, 2 delete, 3 mismatch) to tr[] in push order; rr / cc end at the border.
// CLS: t / qy are class-index strings staged in LDS (lm_plain_path), not the letters in HBM -- the walk is a chain of dependent
// loads (plane words of the next window, the letter under every step), and two of them per window plus one per run were
// HBM / L2 round trips for a byte
template <bool CLS>
__device__ __noinline__ int lm_traceback_planes(const uint32_t* planeH, const uint32_t* planeV, int nw, int nl1, const uint8_t* t,
                                                const uint8_t* qy, int& rr_io, int& cc_io, uint8_t* tr, int lane) {
  // (rr / cc as LOCALS: through the references they live in the caller's frame -- scratch memory -- and every byte store to tr[],
  //  which may alias anything, made the compiler re-load them: two scratch round trips per run, ~2 500 cycles, round 6)
  int tl = 0;
  int rr = rfl(rr_io);
  int cc = rfl(cc_io);
  while (rr > 0 && cc > 0) {
    const int r = rr - lane, c = cc - lane;
    uint32_t wh = 0, wv = 0;
    int rwf = -1;
    uint32_t tmask = 0;   // classes the query letter of this lane's column equals
    if (r >= 1 && c >= 1) {
      const int z = r - 1;
      const int lo = z / (32 * nw);
      const int w = (z >> 5) - lo * nw;
      const size_t wi = ((size_t)(c - 1 + lo) * nw + w) * nl1 + lo;
      rwf = z >> 5;
      wh = ld_scratch(planeH + wi);
      wv = ld_scratch(planeV + wi);
      const int y = CLS ? (int)qy[c - 1] : iupac_index((int)qy[c - 1]);
      tmask = (1u << y) | iupac_partners(y);   // (the relation is symmetric)
    }
    int l = 0;   // columns consumed since the fetch: lane x stands for diagonal offset x - l
    bool inwin = true;
    while (inwin) {
      const int d = lane - l;
      const int rx = rr - d;
      const bool valid = (d >= 0) && (rwf >= 0) && (rx >= 1) && (((rx - 1) >> 5) == rwf);
      uint32_t code = 4u;
      if (valid) {
        const int q = (rx - 1) & 31;
        if ((wh >> q) & 1u) code = (uint32_t)ED_INSERT;
        else if ((wv >> q) & 1u) code = (uint32_t)ED_DELETE;
        else code = ((tmask >> (CLS ? (int)t[rx - 1] : iupac_index((int)t[rx - 1]))) & 1u) ? (uint32_t)ED_MATCH : (uint32_t)ED_MISMATCH;
      }
      const bool isd = (code == (uint32_t)ED_MATCH || code == (uint32_t)ED_MISMATCH);
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
          inwin = false;   // the path left the fetched 32-row word of this column
        } else {
          if (lane == 0) tr[tl] = (uint8_t)cl;
          ++tl;
          if (cl == (uint32_t)ED_INSERT) { --cc; ++l; }   // the column is consumed: next lane
          else --rr;                                        // same column, next row up
          if (rr <= 0 || cc <= 0 || l >= WAVE) inwin = false;
        }
      }
    }
  }
  rr_io = rr;
  cc_io = cc;
  return tl;
}

// row `tlen` of the NW matrix of t (tlen letters) vs q for every column, into row_out[0..qlen]
// (row_out[c] = distance(t, q[0..c))).  bndA / bndB: strip boundary scratch.
__device__ __forceinline__ void lm_last_row(const uint8_t* tp, int tstep, int tlen, const uint8_t* qp, int qstep, int qlen,
                                            int mode, int32_t* bndA, int32_t* bndB, int32_t* row_out, int lane) {
  if ((mode & LM_EQ) && (mode & LM_EQFAST) && tlen >= 1 && qlen >= 1 && tlen <= MYERS_ROWS) {
    if (tlen <= WAVE * 32) lm_last_row_myers<1>(tp, tstep, tlen, qp, qstep, qlen, row_out, lane);
    else if (tlen <= WAVE * 64) lm_last_row_myers<2>(tp, tstep, tlen, qp, qstep, qlen, row_out, lane);
    else lm_last_row_myers<3>(tp, tstep, tlen, qp, qstep, qlen, row_out, lane);
    return;
  }
  const int Q = (tlen + 1 + LRS - 1) / LRS;
  const int pad = Q * LRS - (tlen + 1);   // row tlen = last slot of the last strip
  for (int q = 0; q < Q; ++q) {
    const int32_t* bin = (q > 0) ? ((q & 1) ? bndA : bndB) : nullptr;
    int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bndB : bndA) : row_out;
    (void)lm_pass<false, false>(tp, tstep, tlen, qp, qstep, qlen, q, pad, mode & (LM_EQ | LM_EQFAST), 0, bin, bout, nullptr, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (lane == 0) row_out[0] = tlen;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// the end of a traceback-regime path: border runs (target exhausted -> INSERTs, query exhausted -> DELETEs, edlib.cpp:1027-1091),
// then the ops, walked back to front into tmp[0..tl), appended to `ops` in forward order at `pos`; returns the new position
__device__ __forceinline__ int lm_plain_finish(int tl, int rr, int cc, uint8_t* tmp, uint8_t* ops, int pos, int lane) {
  const int tail = (rr > 0) ? rr : cc;
  const uint8_t op = (rr > 0) ? (uint8_t)ED_DELETE : (uint8_t)ED_INSERT;
  for (int k = lane; k < tail; k += WAVE) tmp[tl + k] = op;
  tl += tail;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k = lane; k < tl; k += WAVE) ops[pos + k] = tmp[tl - 1 - k];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LRT_LAP(9);
  return pos + tl;
}

// plain (traceback-regime) NW path of t[0..tlen) vs q[0..qlen): direction strips + windowed
// run-length traceback; ops appended to `ops` (forward order) at position pos; returns new pos
__device__ __forceinline__ int lm_plain_path(const uint8_t* t, int tlen, const uint8_t* qy, int qlen, int mode, int32_t* bndA,
                                             int32_t* bndB, uint32_t* dirs, uint64_t dirs_cap, uint8_t* tmp,
                                             uint8_t* ops, int pos, int lane) {
  // code words of THIS rectangle: strips of lr_strip_words(qlen) words.  The traceback regime bounds the rectangle
  // (edlib.cpp:1189: < 1 MiB of alignment data), so the direction area is sized by lm_dirs_words() on the host, not by
  // the product of the longest target and the longest query.
  const uint64_t strip_words = lr_strip_words(qlen);
  const int Q = tlen / LRS + 1;
  int rr = tlen, cc = qlen;
  int tl;
  // bit-plane flavour in the compare-free regime, when the two planes fit the room the code words of this
  // rectangle would take (both are 2 bits per cell up to rounding)
  const int bv_nw = (tlen <= WAVE * 32) ? 1 : ((tlen <= WAVE * 64) ? 2 : 3);
  const int bv_nl1 = (tlen >= 1) ? (tlen - 1) / (32 * bv_nw) + 2 : 2;
  const uint64_t bv_words = (uint64_t)(((qlen + bv_nl1 - 2 + 15) >> 4) * 16) * bv_nw * bv_nl1;
  if ((mode & LM_EQ) && (mode & LM_EQFAST) && tlen >= 1 && qlen >= 1 && tlen <= MYERS_ROWS &&
      2 * bv_words <= dirs_cap) {
    uint32_t* planeH = dirs;
    uint32_t* planeV = dirs + bv_words;
    LRT_LAP(12);
    if (bv_nw == 1) lm_dirs_myers<1>(t, tlen, qy, qlen, planeH, planeV, lane);
    else if (bv_nw == 2) lm_dirs_myers<2>(t, tlen, qy, qlen, planeH, planeV, lane);
    else lm_dirs_myers<3>(t, tlen, qy, qlen, planeH, planeV, lane);
    LRT_LAP(8);
    if (tlen + qlen <= MYERS_NW * 16 * WAVE * 4) {
      // the match masks are done with: their LDS holds the two strings as class indices for the walk
      uint8_t* tcl = reinterpret_cast<uint8_t*>(lm_eq_lds());
      uint8_t* qcl = tcl + tlen;
      for (int k = lane; k < tlen; k += WAVE) tcl[k] = (uint8_t)iupac_index((int)t[k]);
      for (int k = lane; k < qlen; k += WAVE) qcl[k] = (uint8_t)iupac_index((int)qy[k]);
      __syncthreads();
      tl = lm_traceback_planes<true>(planeH, planeV, bv_nw, bv_nl1, tcl, qcl, rr, cc, tmp, lane);
      __syncthreads();
    } else {
      tl = lm_traceback_planes<false>(planeH, planeV, bv_nw, bv_nl1, t, qy, rr, cc, tmp, lane);
    }
  } else {
    LRT_LAP(12);
    for (int q = 0; q < Q; ++q) {
      const int32_t* bin = (q > 0) ? ((q & 1) ? bndA : bndB) : nullptr;
      int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bndB : bndA) : nullptr;
      (void)lm_pass<true, false>(t, 1, tlen, qy, 1, qlen, q, 0, mode & (LM_EQ | LM_EQFAST), 0, bin, bout, dirs + (size_t)q * strip_words, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    LRT_LAP(8);
    GeoLR G{dirs, strip_words};
    tl = traceback_runs<true>(G, rr, cc, tmp, lane);
  }
  return lm_plain_finish(tl, rr, cc, tmp, ops, pos, lane);
}

// words of direction codes the traceback-regime rectangles of lm_nw_path can need for targets <= tcap and queries <= qcap:
// max over the query block count of (tl/320 + 1) strips x lr_strip_words(ql), tl the longest target edlib still traces
// directly (edlib.cpp:1185-1191: (2*8 + 4) * blocks * tl + 2*4*tl < 2^20)
inline uint64_t lm_dirs_words(int tcap, int qcap) {
  uint64_t best = (uint64_t)lr_strip_words(64) * 2;
  for (long long blocks = 1; blocks <= (qcap + 63) / 64; ++blocks) {
    long long tl = (1024 * 1024 - 1) / (20 * blocks + 8);
    if (tl > tcap) tl = tcap;
    if (tl < 1) break;
    const int ql = (int)std::min<long long>(blocks * 64, qcap);
    const uint64_t w = (uint64_t)(tl / LRS + 1) * lr_strip_words(ql);
    if (w > best) best = w;
    // the bit-plane flavour of the same rectangle (lm_plain_path): two planes of bv_words
    const long long tb = std::min<long long>(tl, MYERS_ROWS);
    const int nw = (tb <= WAVE * 32) ? 1 : ((tb <= WAVE * 64) ? 2 : 3);
    const long long nl1 = (tb - 1) / (32 * nw) + 2;
    const uint64_t bv = 2ull * (uint64_t)(((ql + nl1 - 2 + 15) >> 4) * 16) * nw * nl1;
    if (bv > best) best = bv;
  }
  return best + 64;
}

// Hirschberg split of a rectangle with query length ql (edlib.cpp:1328-1345): the optimum of left[i] + right[ql - i] and the
// first query index that reaches it (ascending over 1 .. ql - 1, then the two boundary cases)
__device__ __forceinline__ int lm_split_column(const int32_t* left, const int32_t* right, int ql, int lane) {
  int best = 1 << 30;
  for (int i = lane; i <= ql; i += WAVE) best = min(best, left[i] + right[ql - i]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o));
  best = rfl(best);
  int ul = -1;
  for (int base = 1; base <= ql - 1 && ul < 0; base += WAVE) {
    const int i = base + lane;
    const bool hit = (i <= ql - 1) && (left[i] + right[ql - i] == best);
    const unsigned long long bm = __ballot(hit);
    if (bm) ul = base + __builtin_ctzll(bm);
  }
  if (ul < 0) ul = (left[0] + right[ql] == best) ? 0 : ql;
  return rfl(ul);
}

// ---- Hirschberg levels breadth first: every last-row pass of a level in ONE wavefront (round 6) ----------------------------
// The depth-first recursion below runs 2 + 4 + 8 + ... last-row passes, one after the other, each with one column per step
// whatever its row count: a 475-row pass of the third level keeps 15 of the 64 lanes busy, and every level costs as much
// as the first (profiles/r06/lrc_phases_start.txt: 44 % of a msaWfa junction, 41 % of a msaEdlib junction).  But the rectangles
// of one level partition the target, so the rows of ALL their passes together are the rows of the first level's two: they fit
// the lanes of one wavefront side by side.  lm_last_rows_packed runs them as SEGMENTS of lanes -- each with its own target
// slice, direction, query slice and output row -- in one column loop whose length is the level's longest query slice, so
// level k costs 1 / 2^k of the first instead of as much again.  The query's class indices are staged once in LDS (nibbles);
// a lane reads the class of its own column there, so no letter travels through the lanes and a segment can start anywhere.
// The rectangle list lives in LDS in left-to-right order; when every rectangle is in the traceback regime (or the list is
// full) the depth-first code finishes each one in order, so the op string is edlib's byte for byte as before.
constexpr int LM_BFS_RECTS = 32;      // rectangles of a level (a 4 kb x 4 kb alignment ends with 8)
constexpr int LM_QCLS_CAP = 8192;     // query letters the class-index table holds (two per byte)
struct __attribute__((aligned(16))) LmBfsLds {
  int rect[2][LM_BFS_RECTS][4];       // (t0, tl, q0, ql), two generations
  int roff[LM_BFS_RECTS];             // where the rectangle's two score rows start in the left / right row buffers; -1: not split at this level
  unsigned seg[2 * LM_BFS_RECTS];     // segment 2r = left half of rectangle r, 2r + 1 = right half: first lane | lanes << 8 | pass << 16
  uint8_t qcls[LM_QCLS_CAP / 2];
};
__device__ __forceinline__ LmBfsLds& lm_bfs_lds() {
  __shared__ LmBfsLds B;
  return B;
}
__device__ __forceinline__ bool lm_traceback_regime(int tl, int ql) {   // edlib.cpp:1185-1191
  const long long blocks = (ql + 63) / 64;
  return (2ll * 8 + 4) * blocks * tl + 2ll * 4 * tl < 1024 * 1024;
}

// the last-row passes of pass `pass` of the current level (segments assigned by lm_nw_path): for the left half of rectangle
// r, Lrow[roff + i] = distance(t[t0 .. t0 + lw), q[q0 .. q0 + i)); for the right half, Rrow[roff + k] = distance of the
// reversed strings (last k query letters against t[t0 + lw .. t0 + tl)) -- what lm_last_row_myers computes one at a time
template <int NWORDS>
__device__ __noinline__ void lm_last_rows_packed(const uint8_t* target_, int nrect, int cur, int pass, int32_t* Lrow_, int32_t* Rrow_,
                                                 int lane) {
  LmBfsLds& B = lm_bfs_lds();
  uint32_t* E = lm_eq_lds();
  const gptr_cu8 target = (gptr_cu8)target_;
  int myseg = -1, off = 0, nl = 0;
  for (int k = 0; k < 2 * nrect; ++k) {
    const unsigned d = B.seg[k];
    const int fl = (int)(d & 255u), n = (int)((d >> 8) & 255u);
    if ((int)(d >> 16) == pass && lane >= fl && lane < fl + n) { myseg = k; off = lane - fl; nl = n; }
  }
  const bool mine = myseg >= 0;
  const int r = mine ? (myseg >> 1) : 0;
  const bool right = mine && (myseg & 1);
  const int t0 = B.rect[cur][r][0], tl = B.rect[cur][r][1], q0 = B.rect[cur][r][2];
  const int ql = mine ? B.rect[cur][r][3] : 0;
  const int lw = tl / 2, rw = tl - lw;
  const int tlen = mine ? (right ? rw : lw) : 0;
  const int tbase = right ? t0 + tl - 1 : t0, tstep = right ? -1 : 1;
  const int qbase = right ? q0 + ql - 1 : q0, qdir = right ? -1 : 1;
  const gptr_i32 out = (gptr_i32)(right ? Rrow_ : Lrow_) + (mine ? B.roff[r] : 0);
  const bool is_head = off == 0, is_tail = mine && off == nl - 1;
  const int row0 = off * 32 * NWORDS;
  uint32_t mk[NWORDS];   // rows of this lane at or beyond tlen (their vertical deltas are taken off the bottom score)
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    const int lo = row0 + w * 32;
    const int nb = min(32, max(0, lo + 32 - tlen));
    mk[w] = (nb >= 32) ? 0xffffffffu : ((nb > 0) ? (~0u << (32 - nb)) : 0u);
    lm_eq_word(E, w, target + tbase, tstep, lo, tlen, lane);
  }
  int T = mine ? ql + off : 0;   // steps until this lane has seen its last column
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) T = max(T, __shfl_xor(T, o));
  T = rfl(T);
  uint32_t Pv[NWORDS], Mv[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) {
    Pv[w] = 0xffffffffu;
    Mv[w] = 0;
  }
  int score = row0 + 32 * NWORDS;
  int hcarry = 1;
  // class of this lane's column at step t (x WAVE: the slot offset in E); 15 = none outside the segment's columns
  auto load_cls = [&](int t) -> int {
    const int c = t - off;
    const bool in = mine && (unsigned)c < (unsigned)ql;
    const int a = in ? qbase + qdir * c : 0;
    const int nib = ((int)B.qcls[a >> 1] >> ((a & 1) * 4)) & 15;
    return (in ? nib : 15) * WAVE;
  };
  int clsN = load_cls(0), clsNN = load_cls(1);
  uint32_t EqN[NWORDS];
#pragma unroll
  for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + clsN + lane];
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    uint32_t EqC[NWORDS];
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) EqC[w] = EqN[w];
    clsN = clsNN;
    clsNN = load_cls(t + 2);
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) EqN[w] = E[w * 16 * WAVE + clsN + lane];
    int hin = dpp_from_prev(hcarry, 1);
    hin = is_head ? 1 : hin;          // E[0][c] - E[0][c-1] = 1 above a segment's first row
    const int c = t - off;
    const bool valid = mine && (unsigned)c < (unsigned)ql;
    uint32_t nP[NWORDS], nM[NWORDS];
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) {
      uint32_t Eq = EqC[w];
      const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470
      const uint32_t Xv = Eq | Mv[w];
      Eq |= hinNeg;
      const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
      uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
      uint32_t Mh = Pv[w] & Xh;
      const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
      Ph <<= 1;
      Mh <<= 1;
      Mh |= hinNeg;
      Ph |= (hin > 0) ? 1u : 0u;
      nP[w] = Mh | ~(Xv | Ph);
      nM[w] = Ph & Xv;
      hin = hout;
    }
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) {
      Pv[w] = valid ? nP[w] : Pv[w];
      Mv[w] = valid ? nM[w] : Mv[w];
    }
    hcarry = valid ? hin : hcarry;
    score += valid ? hin : 0;
    if (is_tail && valid) {   // E[tlen][c + 1] in the lane that owns row tlen
      int sc = score;
#pragma unroll
      for (int w = 0; w < NWORDS; ++w) sc += __popc(Mv[w] & mk[w]) - __popc(Pv[w] & mk[w]);
      out[c + 1] = sc;
    }
  }
  if (mine && is_head) out[0] = tlen;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// The traceback-regime rectangles a breadth-first run ends with are independent too: the bit-plane fills (lm_dirs_myers) of
// consecutive ones run as segments of one wavefront, each into its own planes inside the direction area; the walks follow one
// after the other.  One word per lane (targets <= 2048 rows each); segment k = rectangle k: B.seg[k] as above (pass = group),
// B.roff[k] = word offset of its planeH (planeV follows at + lm_bv_words).  The query's classes come from B.qcls.
__device__ __forceinline__ uint64_t lm_bv_words1(int tl, int ql) {   // words of ONE plane of a tl x ql rectangle, one word per lane (lm_plain_path)
  const int nl1 = (tl - 1) / 32 + 2;
  return (uint64_t)(((ql + nl1 - 2 + 15) >> 4) * 16) * nl1;
}
__device__ __noinline__ void lm_dirs_packed(const uint8_t* target_, int cur, int r0, int r1, int group, uint32_t* dirs_, int lane) {
  LmBfsLds& B = lm_bfs_lds();
  uint32_t* E = lm_eq_lds();
  const gptr_cu8 target = (gptr_cu8)target_;
  int myseg = -1, off = 0, nl = 0;
  for (int k = r0; k < r1; ++k) {
    const unsigned d = B.seg[k];
    const int fl = (int)(d & 255u), n = (int)((d >> 8) & 255u);
    if ((int)(d >> 16) == group && lane >= fl && lane < fl + n) { myseg = k; off = lane - fl; nl = n; }
  }
  const bool mine = myseg >= 0;
  const int r = mine ? myseg : r0;
  const int t0 = B.rect[cur][r][0], tlen = mine ? B.rect[cur][r][1] : 0, q0 = B.rect[cur][r][2];
  const int ql = mine ? B.rect[cur][r][3] : 0;
  const int nl1 = nl + 1;   // owner lanes + the (unused here) dummy slot of the single-rectangle layout
  const gptr_u32 planeH = (gptr_u32)dirs_ + (mine ? B.roff[r] : 0);
  const gptr_u32 planeV = planeH + (mine ? lm_bv_words1(tlen, ql) : 0);
  const bool is_head = off == 0;
  const int row0 = off * 32;
  lm_eq_word(E, 0, target + t0, 1, row0, tlen, lane);
  int T = mine ? ql + off : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) T = max(T, __shfl_xor(T, o));
  T = rfl(T);
  uint32_t Pv = 0xffffffffu, Mv = 0;
  int hcarry = 1;
  auto load_cls = [&](int t) -> int {
    const int c = t - off;
    const bool in = mine && (unsigned)c < (unsigned)ql;
    const int a = in ? q0 + c : 0;
    const int nib = ((int)B.qcls[a >> 1] >> ((a & 1) * 4)) & 15;
    return (in ? nib : 15) * WAVE;
  };
  int clsN = load_cls(0), clsNN = load_cls(1);
  uint32_t EqN = E[clsN + lane];
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    uint32_t Eq = EqN;
    clsN = clsNN;
    clsNN = load_cls(t + 2);
    EqN = E[clsN + lane];
    int hin = dpp_from_prev(hcarry, 1);
    hin = is_head ? 1 : hin;
    const int c = t - off;
    const bool valid = mine && (unsigned)c < (unsigned)ql;
    const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470
    const uint32_t Xv = Eq | Mv;
    Eq |= hinNeg;
    const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint32_t Ph = Mv | ~(Xh | Pv);
    uint32_t Mh = Pv & Xh;
    const uint32_t PhOut = Ph;                      // horizontal +1 deltas of this column's rows
    const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
    Ph <<= 1;
    Mh <<= 1;
    Mh |= hinNeg;
    Ph |= (hin > 0) ? 1u : 0u;
    const uint32_t nP = Mh | ~(Xv | Ph);
    const uint32_t nM = Ph & Xv;
    if (valid) {   // cell (r, c + 1) of the rectangle: step s = c + off of the single-rectangle layout
      const size_t wi = (size_t)t * nl1 + off;
      planeH[wi] = PhOut;
      planeV[wi] = nP;                              // vertical +1 deltas within this column
    }
    Pv = valid ? nP : Pv;
    Mv = valid ? nM : Mv;
    hcarry = valid ? hout : hcarry;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// edlibAlign(query, target, NW, PATH, extended-IUPAC equalities).alignment of ONE rectangle, depth first (obtainAlignment,
// edlib.cpp:1163-1389); ops appended at `pos`, returns the new position or -1 on overflow.
// (`strip_words` = capacity of `dirs` in words, lm_dirs_words())
__device__ __forceinline__ int lm_nw_dfs(const uint8_t* target, const uint8_t* query, int rt0, int rtl, int rq0, int rql, int mode,
                                         int32_t* bnd, int bnd_stride, uint32_t* dirs, uint64_t strip_words, uint8_t* tmp,
                                         uint8_t* ops, int ops_cap, int pos, int lane) {
  // explicit stack of rectangles (t0, tlen, q0, qlen), processed left to right.  In LDS: indexed by the stack pointer, a local
  // array would live in scratch memory (384 B per lane in every kernel that aligns long strings; one wavefront per workgroup
  // in all of them, and one path at a time)
  __shared__ int st[LM_STACK][4];
  int sp = 0;
  st[sp][0] = rt0; st[sp][1] = rtl; st[sp][2] = rq0; st[sp][3] = rql;
  ++sp;
  int32_t* bndA = bnd;
  int32_t* bndB = bnd + bnd_stride;
  int32_t* left = bnd + 2 * bnd_stride;
  int32_t* right = bnd + 3 * bnd_stride;
  while (sp > 0) {
    --sp;
    const int t0 = rfl(st[sp][0]), tl = rfl(st[sp][1]), q0 = rfl(st[sp][2]), ql = rfl(st[sp][3]);
    if (pos + tl + ql > ops_cap) return -1;
    if (ql == 0 || tl == 0) {   // edlib.cpp:1169-1176
      const uint8_t op = (ql == 0) ? (uint8_t)ED_DELETE : (uint8_t)ED_INSERT;
      for (int k = lane; k < tl + ql; k += WAVE) ops[pos + k] = op;
      pos += tl + ql;
      continue;
    }
    if (lm_traceback_regime(tl, ql)) {
      if ((uint64_t)(tl / LRS + 1) * lr_strip_words(ql) > strip_words) return -1;   // (direction area: sized by lm_dirs_words)
      pos = lm_plain_path(target + t0, tl, query + q0, ql, mode, bndA, bndB, dirs, strip_words, tmp, ops, pos, lane);
      continue;
    }
    // Hirschberg step
    const int lw = tl / 2, rw = tl - lw;
    LRT_LAP(12);
    lm_last_row(target + t0, 1, lw, query + q0, 1, ql, mode, bndA, bndB, left, lane);                        // left[i]  : q[0..i) vs t[0..lw)
    lm_last_row(target + t0 + tl - 1, -1, rw, query + q0 + ql - 1, -1, ql, mode, bndA, bndB, right, lane);   // right[k] : last k letters of q vs t[lw..)
    LRT_LAP(7);
    const int ul = lm_split_column(left, right, ql, lane);
    if (sp + 2 > LM_STACK) return -1;
    st[sp][0] = t0 + lw; st[sp][1] = rw; st[sp][2] = q0 + ul; st[sp][3] = ql - ul;   // lower right (second)
    ++sp;
    st[sp][0] = t0; st[sp][1] = lw; st[sp][2] = q0; st[sp][3] = ul;                  // upper left (first)
    ++sp;
  }
  return pos;
}

// edlibAlign(query, target, NW, PATH, extended-IUPAC equalities).alignment.  Returns the op count, ops[] in forward order;
// -1 on overflow.  Hirschberg levels breadth first with packed last-row passes while the compare-free regime and the LDS
// tables allow (see above), then depth first from every rectangle of the list.
__device__ __forceinline__ int lm_nw_path(const uint8_t* target, int tn, const uint8_t* query, int qn, int mode, int32_t* bnd,
                                          int bnd_stride, uint32_t* dirs, uint64_t strip_words, uint8_t* tmp,
                                          uint8_t* ops, int ops_cap, int lane) {
  LmBfsLds& B = lm_bfs_lds();
  int cur = 0, nrect = 1;
  if (lane == 0) { B.rect[0][0][0] = 0; B.rect[0][0][1] = tn; B.rect[0][0][2] = 0; B.rect[0][0][3] = qn; }
  int32_t* left = bnd + 2 * bnd_stride;
  int32_t* right = bnd + 3 * bnd_stride;
  const bool packed = (mode & LM_EQ) && (mode & LM_EQFAST) && qn >= 1 && tn >= 2 && qn <= LM_QCLS_CAP && tn <= 2 * MYERS_ROWS &&
                      qn + LM_BFS_RECTS < bnd_stride && !lm_traceback_regime(tn, qn);
  if (packed) {
    LRT_LAP(12);
    for (int i = lane; i < (qn + 1) / 2; i += WAVE) {   // (compare-free regime: every letter is one of the 15 classes)
      const int lo = iupac_index((int)query[2 * i]);
      const int hi = (2 * i + 1 < qn) ? iupac_index((int)query[2 * i + 1]) : 15;
      B.qcls[i] = (uint8_t)((lo & 15) | ((hi & 15) << 4));
    }
    __syncthreads();
    for (;;) {
      int t0 = 0, tl = 0, q0 = 0, ql = 0;
      if (lane < nrect) { t0 = B.rect[cur][lane][0]; tl = B.rect[cur][lane][1]; q0 = B.rect[cur][lane][2]; ql = B.rect[cur][lane][3]; }
      const bool split = lane < nrect && ql > 0 && tl >= 2 && !lm_traceback_regime(tl, ql);
      const unsigned long long sm = __ballot(split);
      const int nsplit = __popcll(sm);
      if (nsplit == 0 || nrect + nsplit > LM_BFS_RECTS) break;
      const int lw = tl / 2, rw = tl - lw;
      int nw = 3;   // words per lane: the smallest that seats every segment of the level in one pass
      {
        int need1 = split ? (lw + 31) / 32 + (rw + 31) / 32 : 0, need2 = split ? (lw + 63) / 64 + (rw + 63) / 64 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { need1 += __shfl_xor(need1, o); need2 += __shfl_xor(need2, o); }
        nw = (rfl(need1) <= WAVE) ? 1 : ((rfl(need2) <= WAVE) ? 2 : 3);
      }
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      if (lane < nrect) {
        B.roff[lane] = split ? q0 + __popcll(sm & below) : -1;
        B.seg[2 * lane] = split ? (unsigned)((lw + 32 * nw - 1) / (32 * nw)) : 0u;
        B.seg[2 * lane + 1] = split ? (unsigned)((rw + 32 * nw - 1) / (32 * nw)) : 0u;
      }
      __syncthreads();
      int npass = 0, used = 0;
      for (int k = 0; k < 2 * nrect; ++k) {   // greedy seating, left to right (uniform)
        const int n = rfl((int)B.seg[k]);
        unsigned d = 0xffffffffu;
        if (n > 0) {
          if (used + n > WAVE) { ++npass; used = 0; }
          d = (unsigned)used | ((unsigned)n << 8) | ((unsigned)npass << 16);
          used += n;
        }
        if (lane == 0) B.seg[k] = d;
      }
      npass += 1;
      __syncthreads();
      for (int p = 0; p < npass; ++p) {
        if (nw == 1) lm_last_rows_packed<1>(target, nrect, cur, p, left, right, lane);
        else if (nw == 2) lm_last_rows_packed<2>(target, nrect, cur, p, left, right, lane);
        else lm_last_rows_packed<3>(target, nrect, cur, p, left, right, lane);
      }
      LRT_LAP(7);
      // next generation, in order: a rectangle that was not split stays, a split one becomes its two halves
      int at = 0;
      for (int r = 0; r < nrect; ++r) {
        const int rt0 = rfl(B.rect[cur][r][0]), rtl = rfl(B.rect[cur][r][1]), rq0 = rfl(B.rect[cur][r][2]), rql = rfl(B.rect[cur][r][3]);
        const int ro = rfl(B.roff[r]);
        if (ro < 0) {
          if (lane == 0) { B.rect[cur ^ 1][at][0] = rt0; B.rect[cur ^ 1][at][1] = rtl; B.rect[cur ^ 1][at][2] = rq0; B.rect[cur ^ 1][at][3] = rql; }
          at += 1;
        } else {
          const int ul = lm_split_column(left + ro, right + ro, rql, lane);
          const int rlw = rtl / 2;
          if (lane == 0) {
            B.rect[cur ^ 1][at][0] = rt0; B.rect[cur ^ 1][at][1] = rlw; B.rect[cur ^ 1][at][2] = rq0; B.rect[cur ^ 1][at][3] = ul;
            B.rect[cur ^ 1][at + 1][0] = rt0 + rlw; B.rect[cur ^ 1][at + 1][1] = rtl - rlw; B.rect[cur ^ 1][at + 1][2] = rq0 + ul; B.rect[cur ^ 1][at + 1][3] = rql - ul;
          }
          at += 2;
        }
      }
      __syncthreads();
      cur ^= 1;
      nrect = at;
      LRT_LAP(12);
    }
  }
  __syncthreads();
  int pos = 0;
  int r = 0, group = 0;
  while (r < nrect && pos >= 0) {
    // consecutive traceback-regime rectangles whose plane fills fit one wavefront and the direction area together
    int e = r, lanes = 0, need = 0;
    uint64_t words = 0;
    if (packed && nrect > 1) {
      for (; e < nrect; ++e) {
        const int tl = rfl(B.rect[cur][e][1]), ql = rfl(B.rect[cur][e][3]);
        if (ql <= 0 || tl <= 0 || tl > WAVE * 32 || !lm_traceback_regime(tl, ql)) break;
        const int n = (tl + 31) / 32;
        const uint64_t w = 2 * lm_bv_words1(tl, ql);
        if (lanes + n > WAVE || words + w > strip_words || words + w > 0x7fffffffull) break;
        if (lane == 0) { B.seg[e] = (unsigned)lanes | ((unsigned)n << 8) | ((unsigned)group << 16); B.roff[e] = (int)words; }
        lanes += n;
        words += w;
        need += tl + ql;
      }
    }
    if (e - r >= 2) {
      if (pos + need > ops_cap) return -1;
      __syncthreads();
      LRT_LAP(12);
      lm_dirs_packed(target, cur, r, e, group, dirs, lane);
      LRT_LAP(8);
      for (int k = r; k < e; ++k) {
        const int rt0 = rfl(B.rect[cur][k][0]), rtl = rfl(B.rect[cur][k][1]), rq0 = rfl(B.rect[cur][k][2]), rql = rfl(B.rect[cur][k][3]);
        const uint32_t* planeH = dirs + rfl(B.roff[k]);
        const uint32_t* planeV = planeH + lm_bv_words1(rtl, rql);
        uint8_t* tcl = reinterpret_cast<uint8_t*>(lm_eq_lds());   // (the match masks are done with)
        uint8_t* qcl = tcl + rtl;
        for (int i = lane; i < rtl; i += WAVE) tcl[i] = (uint8_t)iupac_index((int)target[rt0 + i]);
        for (int i = lane; i < rql; i += WAVE) qcl[i] = (uint8_t)iupac_index((int)query[rq0 + i]);
        __syncthreads();
        int rr = rtl, cc = rql;
        const int tl = lm_traceback_planes<true>(planeH, planeV, 1, (rtl - 1) / 32 + 2, tcl, qcl, rr, cc, tmp, lane);
        __syncthreads();
        pos = lm_plain_finish(tl, rr, cc, tmp, ops, pos, lane);
      }
      group += 1;
      r = e;
    } else {
      const int rt0 = rfl(B.rect[cur][r][0]), rtl = rfl(B.rect[cur][r][1]), rq0 = rfl(B.rect[cur][r][2]), rql = rfl(B.rect[cur][r][3]);
      pos = lm_nw_dfs(target, query, rt0, rtl, rq0, rql, mode, bnd, bnd_stride, dirs, strip_words, tmp, ops, ops_cap, pos, lane);
      r += 1;
    }
  }
  return pos;
}

// ---- edlibAlign HW / SHW on long strings (splitAlign of long-read insertions, msaWfa) ---------
struct LmRes {
  int ed, endLoc, startLoc, nops;   // ops (forward order) in the caller's buffer
};

__device__ __forceinline__ int lm_first_row(int qn) { return ((qn & 63) != 0) ? 0 : 1; }   // see ed_first_row (ins_kernel.hpp)

// distance + end location over all target rows (strips); hw: LM_HW or 0 (SHW).  which = 0: first optimal end, 1: last
__device__ __forceinline__ void lm_locate(const uint8_t* tp, int tstep, int tn, const uint8_t* qp, int qstep, int qn, int mode,
                                          int32_t* bndA, int32_t* bndB, int lane, int& ed, int& first, int& last,
                                          int32_t* lastcol = nullptr) {
  const int Q = tn / LRS + 1;
  const int r0 = lm_first_row(qn);
  unsigned kf = 0xffffffffu, kl = 0xffffffffu;
  if (!lastcol && (mode & LM_EQ) && (mode & LM_EQFAST) && tn >= 1 && qn >= 1 && tn <= MYERS_ROWS) {
    const bool hw = (mode & LM_HW) != 0;
    LmKeys k;
    if (tn <= WAVE * 32) k = lm_locate_myers<1>(tp, tstep, tn, qp, qstep, qn, hw, r0, lane);
    else if (tn <= WAVE * 64) k = lm_locate_myers<2>(tp, tstep, tn, qp, qstep, qn, hw, r0, lane);
    else k = lm_locate_myers<3>(tp, tstep, tn, qp, qstep, qn, hw, r0, lane);
    ed = (int)(k.kf >> LM_RBITS);
    first = (int)(k.kf & (unsigned)LM_RMASK);
    last = LM_RMASK - (int)(k.kl & (unsigned)LM_RMASK);
    return;
  }
  for (int q = 0; q < Q; ++q) {
    const int32_t* bin = (q > 0) ? ((q & 1) ? bndA : bndB) : nullptr;
    int32_t* bout = (q + 1 < Q) ? ((q & 1) ? bndB : bndA) : nullptr;
    const LmKeys k = lm_pass<false, true>(tp, tstep, tn, qp, qstep, qn, q, 0, mode, r0, bin, bout, nullptr, lane, lastcol);
    kf = min(kf, k.kf);
    kl = min(kl, k.kl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  ed = (int)(kf >> LM_RBITS);
  first = (int)(kf & (unsigned)LM_RMASK);
  last = LM_RMASK - (int)(kl & (unsigned)LM_RMASK);
}

__device__ __forceinline__ int lm_fill_inserts(uint8_t* ops, int qn, int lane) {
  for (int k = lane; k < qn; k += WAVE) ops[k] = (uint8_t)ED_INSERT;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return qn;
}

// edlibAlign(Q, T, HW, {DISTANCE | LOC | PATH}) (edlib.cpp:139-300); tn >= 1, qn >= 1, tn <= LM_RMASK - 1
__device__ __forceinline__ LmRes lm_hw(const uint8_t* T, int tn, const uint8_t* Qy, int qn, int mode, bool loc, bool path,
                                       int32_t* bnd, int bnd_stride, uint32_t* dirs, uint64_t strip_words, uint8_t* tmp,
                                       uint8_t* ops, int ops_cap, int lane) {
  LmRes o;
  int first, last;
  LRT_LAP(12);
  lm_locate(T, 1, tn, Qy, 1, qn, (mode & (LM_EQ | LM_EQFAST)) | LM_HW, bnd, bnd + bnd_stride, lane, o.ed, first, last);
  LRT_LAP(5);
  o.endLoc = first - 1;
  o.startLoc = 0;
  o.nops = 0;
  if (!loc) return o;
  if (o.endLoc == -1) {   // edlib.cpp:222-235
    if (path) o.nops = lm_fill_inserts(ops, qn, lane);
    return o;
  }
  int ed2, f2, l2;
  lm_locate(T + o.endLoc, -1, o.endLoc + 1, Qy + (qn - 1), -1, qn, mode & (LM_EQ | LM_EQFAST), bnd, bnd + bnd_stride, lane, ed2, f2, l2);
  LRT_LAP(6);
  o.startLoc = o.endLoc - (l2 - 1);
  if (!path) return o;
  const int tl2 = o.endLoc - o.startLoc + 1;
  if (tl2 <= 0) {
    o.nops = lm_fill_inserts(ops, qn, lane);
    return o;
  }
  o.nops = lm_nw_path(T + o.startLoc, tl2, Qy, qn, mode & (LM_EQ | LM_EQFAST), bnd, bnd_stride, dirs, strip_words, tmp, ops, ops_cap, lane);
  return o;
}

// edlibAlign(Q, T, SHW, PATH)
__device__ __forceinline__ LmRes lm_shw(const uint8_t* T, int tn, const uint8_t* Qy, int qn, int mode, int32_t* bnd,
                                        int bnd_stride, uint32_t* dirs, uint64_t strip_words, uint8_t* tmp, uint8_t* ops,
                                        int ops_cap, int lane) {
  LmRes o;
  int first, last;
  lm_locate(T, 1, tn, Qy, 1, qn, mode & (LM_EQ | LM_EQFAST), bnd, bnd + bnd_stride, lane, o.ed, first, last);
  o.endLoc = first - 1;
  o.startLoc = 0;
  if (o.endLoc == -1) o.nops = lm_fill_inserts(ops, qn, lane);
  else o.nops = lm_nw_path(T, o.endLoc + 1, Qy, qn, mode & (LM_EQ | LM_EQFAST), bnd, bnd_stride, dirs, strip_words, tmp, ops, ops_cap, lane);
  return o;
}

// msaEdlib for one junction
__device__ void lrmsa_junction(const LrMsaArgs& A, int j, LrMsaLds& L, uint8_t* ws, int lane) {
  const dellyhip_junction J = A.junc[j];
  dellyhip_result* out = &A.res[j];
  uint8_t* cons_out = A.out_blob + (size_t)j * A.out_stride;
  const int N = J.n_seq;
  if (J.svt == 4) return;   // insertions: msaWfa (lrwfa_kernel.hpp)
  int status = 0, cons_len = 0, rows = 0;
#ifdef DH_LR_TIMING
  const unsigned long long lrt_j0 = wall_clock64();
  LRT_START();
#endif
  uint8_t* alnA = ws;
  uint8_t* alnB = ws + A.off_alnB;
  uint8_t* astr = ws + A.off_astr;
  int32_t* bnd = reinterpret_cast<int32_t*>(ws + A.off_bnd);
  uint8_t* ops = ws + A.off_ops;
  uint8_t* tmp = ws + A.off_tmp;
  uint8_t* cbuf = ws + A.off_cons;
  uint32_t* dirs = reinterpret_cast<uint32_t*>(ws + A.off_dirs);
  const int bnd_stride = A.ncap + 128;
  const int acap = A.acap;
  if (N >= 1) {
    if (N > LM_NR) status = DELLYHIP_E_LIMIT;
    if (!status) {
      for (int r = lane; r < N; r += WAVE) {
        const uint64_t a = A.seq_off[J.seq_first + r], b = A.seq_off[J.seq_first + r + 1];
        L.roff[r] = a;
        L.rlen[r] = (int32_t)(b - a);
      }
      __syncthreads();
      for (int r = 0; r < N; ++r)
        if (L.rlen[r] > A.ncap || L.rlen[r] < 1) status = DELLYHIP_E_LIMIT;
    }
    const int32_t* E = A.edit + (size_t)j * LM_NR * LM_NR;
    if (!status) {
      for (int q = lane; q < N * N; q += WAVE) {
        const int a = q / N, b = q - a * N;
        if (a != b && E[a * LM_NR + b] < 0) status = DELLYHIP_E_LIMIT;
      }
      status = (__ballot(status != 0) != 0ull) ? DELLYHIP_E_LIMIT : 0;
    }
    if (!status) {
      // ---- medoid (assemble.h:397-408): median = element of rank N/2 of each row (diagonal = 0)
      if (lane < N) {
        int med = 0;
        for (int x = 0; x < N; ++x) {
          const int vx = (x == lane) ? 0 : E[lane * LM_NR + x];
          int rank = 0;
          for (int y = 0; y < N; ++y) {
            const int vy = (y == lane) ? 0 : E[lane * LM_NR + y];
            rank += (vy < vx || (vy == vx && y < x)) ? 1 : 0;
          }
          if (rank == N / 2) med = vx;
        }
        L.med[lane] = med;
      }
      __syncthreads();
      int bestIdx = 0, bestVal = L.rlen[0];
      for (int i = 0; i < N; ++i)
        if (L.med[i] < bestVal) { bestVal = L.med[i]; bestIdx = i; }
      // ---- order by (distance to the medoid, index), medoid first; keep 80 % (>= 3) (:410-424)
      uint32_t lastIdx = (uint32_t)(0.8 * N);
      if (lastIdx < 3) lastIdx = 3;
      const int nsel = min((int)lastIdx, N);
      if (lane < N) {
        const int kx = (lane == bestIdx) ? 0 : E[bestIdx * LM_NR + lane];
        int rank = 0;
        for (int y = 0; y < N; ++y) {
          const int ky = (y == bestIdx) ? 0 : E[bestIdx * LM_NR + y];
          // pair order (score, index); the medoid's pair is (0, bestIdx)
          rank += (ky < kx || (ky == kx && y < lane)) ? 1 : 0;
        }
        if (rank < LM_NR) L.sel[rank] = lane;
      }
      __syncthreads();
      // ---- progressive alignment (:426-447)
      const uint8_t* blob = A.seq_blob;
      uint8_t* cur = alnA;
      uint8_t* nxt = alnB;
      int arows = 1, acols = L.rlen[L.sel[0]];
      if (acols > acap - 1) status = DELLYHIP_E_LIMIT;
      if (!status) {
        const uint8_t* r0 = blob + L.roff[L.sel[0]];
        for (int k = lane; k < acols; k += WAVE) cur[k] = r0[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      for (int step = 1; step < nsel && !status; ++step) {
        // consensusEdlib (:198-259)
        for (int col = lane; col < acols; col += WAVE) {
          // (the five counts as bytes of one word, the letter's slot from a nibble table: an if-else chain over per-lane letters is a tree of
          //  divergent branches; rows <= 255)
          unsigned long long packed = 0ull;
          for (int r = 0; r < arows; ++r) {
            const uint8_t ch = cur[(size_t)r * acap + col];
            const int v = letter_code_bf((uint8_t)(ch & 0xDF));   // A/a 0, C/c 1, G/g 2, T/t 3; N/n, '-' and everything else: slot 4
            packed += 1ull << (8 * ((v < 0) ? 4 : v));
          }
          int count[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) count[i] = (int)((packed >> (8 * i)) & 0xffull);
          int maxIdx = 0, sndIdx = 1;
          if (count[maxIdx] < count[sndIdx]) { maxIdx = 1; sndIdx = 0; }
#pragma unroll
          for (int i = 2; i < 5; ++i) {
            if (count[i] > count[maxIdx]) { sndIdx = maxIdx; maxIdx = i; }
            else if (count[i] > count[sndIdx]) sndIdx = i;
          }
          uint8_t letter;
          if (2 * count[sndIdx] < count[maxIdx]) letter = (maxIdx < 4) ? (uint8_t)("ACGT"[maxIdx]) : (uint8_t)'-';
          else {
            const int k1 = min(maxIdx, sndIdx), k2 = max(maxIdx, sndIdx);
            // ACGT- pairs: M R W B / S Y D / K E / F
            const int code = k1 * 5 + k2;
            letter = code == 1 ? 'M' : code == 2 ? 'R' : code == 3 ? 'W' : code == 4 ? 'B' : code == 7 ? 'S' : code == 8 ? 'Y'
                   : code == 9 ? 'D' : code == 13 ? 'K' : code == 14 ? 'E' : code == 19 ? 'F' : '-';
          }
          astr[col] = letter;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LRT_LAP(4);
        const int rd = L.sel[step];
        const uint8_t* qy = blob + L.roff[rd];
        const int qn = L.rlen[rd];
        const int eqmode = LM_EQ | ((lm_in_classes(astr, 1, acols, lane) && lm_in_classes(qy, 1, qn, lane)) ? LM_EQFAST : 0);
#ifdef DH_LR_TIMING
        const unsigned long long lrt_p0 = wall_clock64();
#endif
        const int nops = lm_nw_path(astr, acols, qy, qn, eqmode, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, acap + A.ncap, lane);
#ifdef DH_LR_TIMING
        LRT_ADD(13, wall_clock64() - lrt_p0);
        LRT_LAP(12);
#endif
        if (nops < 0 || nops > acap - 1) { status = DELLYHIP_E_LIMIT; break; }   // (the next target must fit the strips)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // convertAlignment(query, align, NW, cigar) (:24-88)
        int tbase = 0, qbase = 0;
        for (int base = 0; base < nops; base += WAVE) {
          const int jc = base + lane;
          const int op = (jc < nops) ? (int)ops[jc] : ED_MATCH;
          const unsigned long long mt = __ballot(jc < nops && op != ED_INSERT);
          const unsigned long long mq = __ballot(jc < nops && op != ED_DELETE);
          const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int ti = tbase + __popcll(mt & below), qi = qbase + __popcll(mq & below);
          if (jc < nops) {
            for (int r = 0; r < arows; ++r) nxt[(size_t)r * acap + jc] = (op != ED_INSERT) ? cur[(size_t)r * acap + ti] : (uint8_t)'-';
            nxt[(size_t)arows * acap + jc] = (op != ED_DELETE) ? qy[qi] : (uint8_t)'-';
          }
          tbase += __popcll(mt);
          qbase += __popcll(mq);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LRT_LAP(10);
        uint8_t* sw = cur; cur = nxt; nxt = sw;
        arows += 1;
        acols = nops;
      }
      if (!status) {
        // consensus (src/msa.h:111-173), then trim 5 % per side, at most 50 (:465-469)
        Node nd{cur, arows, acols, acap};
        int Lc = consensus_node(nd, A.p, cbuf, acap, L, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int trim = (int)(0.05 * Lc);
        if (trim > 50) trim = 50;
        const int len = Lc - 2 * trim;
        int o = 0;
        if (len > 100) { o = trim; Lc = len; }
        if (Lc > A.out_cons_cap) status = DELLYHIP_E_LIMIT;
        else {
          for (int k = lane; k < Lc; k += WAVE) cons_out[k] = cbuf[o + k];
          cons_len = Lc;
          rows = arows;
        }
      }
    }
  }
#ifdef DH_LR_TIMING
  LRT_LAP(11);
  LRT_ADD(14, 1);
  LRT_ADD(15, wall_clock64() - lrt_j0);
#endif
  if (lane == 0) {
    out->sr_support = rows;
    out->status = status;
    A.cons_len[j] = status ? 0 : cons_len;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__global__ __launch_bounds__(WAVE) void lrmsa_kernel(LrMsaArgs A) {
  __shared__ LrMsaLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) lrmsa_junction(A, w, L, ws, lane);
}

}  // namespace dh
