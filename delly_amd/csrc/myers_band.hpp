// myers_band.hpp -- banded NW edit distance, SEVERAL pairs per wavefront (round 6).
//
// The all-pairs stages of the long-read consensus (msaEdlib's distance matrix src/assemble.h:386-395, msaWfa's trimmed
// distances :551-574) and the genotyper's _editDistanceNW (src/genotype.h:21-30) need the DISTANCE only -- a unique
// number, so any exact method gives edlib's answer.  The full-matrix passes of myers_kernel.hpp compute every cell of a
// 2.2 kb x 2.2 kb matrix for reads whose distance is ~260: a path of cost <= k never leaves the diagonals |row - column| <= k
// (every step off the diagonal costs one), i.e. 27 % of the cells at k = 300.  This is Ukkonen's band on Myers / Hyyro's
// block recurrence (what edlib itself does, src/edlib.cpp:545-700, with a dynamic band), arranged for a 64-lane wavefront:
//
//   * a pair owns WL consecutive lanes; lane i of a pair holds one BLOCK of 64 pattern rows (two 32-bit words), and the
//     WL blocks form a window that slides down the pattern: block b needs the text columns [64 b - k, 64 b + 63 + k] only;
//   * block b processes column it - b at iteration `it` (one column of skew per block, the horizontal delta leaving a block
//     reaches the block below through one DPP move, as in the full passes);
//   * when the top block of the window has seen its last in-band column, every block moves up one lane (its bit-vectors, score
//     and pending delta with it -- DPP moves) and the bottom lane starts a fresh block (vertical deltas all +1, the assumption
//     of the band's lower edge, edlib.cpp:640-655); the block indices move with the window, so no block skips a column;
//   * the top block of the window takes +1 as its incoming delta (the band's upper edge, edlib.cpp:604);
//   * WL >= (64 + 2k) / 65 guarantees that a block enters the window before its first in-band column.
//   With k = 15.6 % of the longer string + 32, WL = 12 lanes for 2.2 kb reads: FIVE pairs per wavefront and step, against two
//   (myers_nw_fast_x2) -- 36 k instead of 123 k wave-instructions per pair.
//   The result is exact iff it is <= k (the optimal path then lies inside the band, where every cell has its true value; cells
//   at the band's edges are over-estimates); otherwise -- or for strings with letters outside A, C, G, T, N, or whose lengths
//   differ by more than k -- the caller runs the full pass.  Match masks (per block and letter) and the text (2-bit codes)
//   of all pairs are staged in LDS: 10 KB per wavefront, so that fifteen fit a CU.
#pragma once

namespace dh {

constexpr int MB_NW = 2;               // words per block
constexpr int MB_BH = 32 * MB_NW;      // rows per block
constexpr int MB_G = 8;                // pairs per wavefront at most
constexpr int MB_EQW = 1600;           // words of match masks per wavefront (A, C, G, T x 2 words per block)
constexpr int MB_TXT = 3072;           // bytes of packed text codes per wavefront (four letters per byte)

struct MbItem {
  const uint8_t* pat;   // rows (the shorter string)
  const uint8_t* txt;   // columns
  int pn, tn;
};
struct __attribute__((aligned(16))) MbLds {
  uint32_t eq[MB_EQW];
  uint8_t txt[MB_TXT];
  MbItem item[MB_G];
  int32_t eqoff[MB_G], txoff[MB_G], res[MB_G], bad[MB_G];
};

__host__ __device__ inline int mb_band(int len) { return (len * 5) / 32 + 32; }                                       // k for strings up to len
__host__ __device__ inline int mb_lanes(int k) { return (MB_BH + 2 * k + MB_BH) / (MB_BH + 1); }                      // ceil((64 + 2k) / 65)
__host__ __device__ inline int mb_eq_words(int pn) { return ((pn + MB_BH - 1) / MB_BH) * 4 * MB_NW; }
__host__ __device__ inline int mb_txt_bytes(int tn) { return ((tn + 3) / 4 + 3) & ~3; }
// pairs per wavefront for strings up to maxlen (0: the band does not pay -- windows of more than half a wavefront)
__host__ __device__ inline int mb_group(int maxlen) {
  const int wl = mb_lanes(mb_band(maxlen));
  int g = WAVE / wl;
  if (g > MB_G) g = MB_G;
  const int by_eq = MB_EQW / (mb_eq_words(maxlen) > 0 ? mb_eq_words(maxlen) : 1);
  const int by_txt = MB_TXT / mb_txt_bytes(maxlen);
  if (g > by_eq) g = by_eq;
  if (g > by_txt) g = by_txt;
  return g >= 2 ? g : 0;
}

// LDS of the kernels that run both flavours: the letter table, then the full passes' match masks and the banded passes'
// tables in the same bytes (a wavefront runs one at a time; the band's results are copied out before a full pass runs).
// Reached through this accessor, not through a reference argument (a reference argument of a called function is a generic
// pointer: every access would be a flat_* instruction).
struct __attribute__((aligned(16))) MyersBandLds {
  uint16_t lut[256];
  unsigned char rest[sizeof(MbLds) > sizeof(MyersLds<MYERS_NW>) - 512 ? sizeof(MbLds) : sizeof(MyersLds<MYERS_NW>) - 512];
  __device__ __forceinline__ MyersLds<MYERS_NW>& full() { return *reinterpret_cast<MyersLds<MYERS_NW>*>(this); }
  __device__ __forceinline__ MbLds& band() { return *reinterpret_cast<MbLds*>(rest); }
};
static_assert(offsetof(MyersLds<MYERS_NW>, eq) == 512, "the letter table is the first 512 bytes of MyersLds");
__device__ __forceinline__ MyersBandLds& myers_band_lds() {
  __shared__ MyersBandLds S;
  return S;
}

// G pairs (M.item[0 .. G)), pair g on lanes [g * wl, (g + 1) * wl); k, wl uniform.  M.res[g] = the distance, or -1 when it is
// not certified (beyond k, a letter outside A, C, G, T, lengths too far apart, tables too small): the caller falls back for that pair.
__device__ __noinline__ void myers_band_multi(int G, int k, int wl, int lane) {
  MbLds& M = myers_band_lds().band();
  const uint16_t* lut = myers_band_lds().lut;
  const int g = lane / wl;
  const bool mine = g < G;
  const int i = lane - g * wl;
  // ---- table offsets, eligibility (uniform per pair, computed by every lane)
  if (lane < G) {
    int eo = 0, to = 0;
    for (int q = 0; q < lane; ++q) {
      eo += mb_eq_words(M.item[q].pn);
      to += mb_txt_bytes(M.item[q].tn);
    }
    const MbItem I = M.item[lane];
    const int ew = mb_eq_words(I.pn), tb = mb_txt_bytes(I.tn);
    const bool fits = eo + ew <= MB_EQW && to + tb <= MB_TXT;
    const bool ok = I.pn >= 1 && I.tn >= 1 && I.pn <= I.tn + k && I.tn <= I.pn + k && fits;
    M.eqoff[lane] = eo;
    M.txoff[lane] = to;
    M.bad[lane] = ok ? 0 : 1;
    M.res[lane] = -1;
  }
  DH_SYNC();
  const MbItem I = M.item[mine ? g : 0];
  const bool live = mine && M.bad[mine ? g : 0] == 0;
  const int eqoff = M.eqoff[mine ? g : 0], txoff = M.txoff[mine ? g : 0];
  const int pn = live ? I.pn : 0, tn = live ? I.tn : 0;
  const gptr_cu8 pat = (gptr_cu8)I.pat, txt = (gptr_cu8)I.txt;
  const int nb = (pn + MB_BH - 1) / MB_BH;       // blocks of the pattern
  const int blast = nb - 1;
  // ---- match masks: lane i builds the blocks i, i + wl, ... of its pair; slot 5 stays zero.  The strings are read sixteen / eight
  //      bytes at a time whatever their length: the sequence blob is padded (batch_upload_impl), a byte beyond the string is ignored
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_u;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef u32x2 __attribute__((aligned(1))) u32x2_u;
  int foreign = 0;
  for (int b = i; b < nb; b += wl) {
    uint32_t* slot = &M.eq[eqoff + b * 4 * MB_NW];
#pragma unroll
    for (int q = 0; q < 4 * MB_NW; ++q) slot[q] = 0;
    const int r0 = b * MB_BH;
    u32x4 v[MB_BH / 16];
#pragma unroll
    for (int q = 0; q < MB_BH / 16; ++q) v[q] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_u*>(pat + r0 + 16 * q);
#pragma unroll 8
    for (int q = 0; q < MB_BH; ++q) {
      const uint32_t wd = v[q >> 4][(q >> 2) & 3];
      const int code = (int)lut[(wd >> ((q & 3) * 8)) & 0xff] >> 6;     // (lut holds slot * WAVE: A 0, C 1, G 2, T 3, N 4, others 5)
      const bool in = r0 + q < pn;
      foreign |= in && code >= 4;                                        // (N and everything else: the full pass compares bytes)
      if (in && code < 4) slot[code * MB_NW + (q >> 5)] |= 1u << (q & 31);
    }
  }
  // ---- the text as 2-bit codes: lane i packs the letters [16 i, 16 i + 16), [16 (i + wl), ...) of its pair
  for (int p0 = i * 16; p0 < tn; p0 += wl * 16) {
    const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_u*>(txt + p0);
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ch = (int)((v[q >> 2] >> ((q & 3) * 8)) & 0xff);
      const bool in = p0 + q < tn;
      const int code = in ? ((int)lut[ch] >> 6) : 0;
      foreign |= in && code >= 4;
      packed |= (uint32_t)(code & 3) << (2 * q);
    }
    *reinterpret_cast<uint32_t*>(&M.txt[txoff + (p0 >> 2)]) = packed;
  }
  if (live && foreign) M.bad[g] = 1;     // (any lane of the pair: same value)
  DH_SYNC();
  const bool run = live && M.bad[mine ? g : 0] == 0;
  // ---- the band.  Block b processes column it - b at iteration `it` whatever the window does (lane i of a window whose top block
  //      is p holds block p + i; a slide adds one to every block index of the pair and the iteration counter goes on, so the
  //      column stays): the horizontal delta a block needs always left the block above in the iteration before.  The slides
  //      follow a fixed schedule -- the first when the top block has seen column 63 + k, then every 65 iterations -- which is
  //      the same for every pair of the wavefront (a scalar countdown) until a pair's window has reached its last block.
  uint32_t Pv[MB_NW], Mv[MB_NW];
#pragma unroll
  for (int w = 0; w < MB_NW; ++w) { Pv[w] = 0xffffffffu; Mv[w] = 0; }
  int hcarry = 1;
  int score = (i + 1) * MB_BH;       // D[last row of the block][0]
  int bcur = i;                      // this lane's block
  int slides_left = run ? max(0, blast - (wl - 1)) : 0;   // (uniform per pair)
  const bool bottom = i == wl - 1, top = i == 0;
  int T = run ? tn + blast : 0;      // iterations until the pair's last block has seen the last column
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) T = max(T, __shfl_xor(T, o));
  T = rfl(T);
  // the code of a text column and the masks of (block, code), fetched one iteration ahead; the code of the column after next
  // travels in codeN, so that the mask address of the next iteration depends on registers only
  auto load_code = [&](int col) -> int {
    const int ca = ((unsigned)col < (unsigned)tn) ? col : 0;
    return ((int)M.txt[txoff + (ca >> 2)] >> ((ca & 3) * 2)) & 3;
  };
  int c = -i;                        // column of the coming iteration
  int codeC = load_code(c), codeN = load_code(c + 1);
  uint32_t Eq[MB_NW];
  {
    const uint32_t* eqp = &M.eq[eqoff + (min(bcur, max(blast, 0)) * 4 + codeC) * MB_NW];
#pragma unroll
    for (int w = 0; w < MB_NW; ++w) Eq[w] = eqp[w];
  }
  int countdown = MB_BH + k;         // iterations until the first slide
  for (int it = 0; it < T; ++it) {
    // ---- this iteration's column with the masks fetched during the last one
    const bool valid = run && (unsigned)c < (unsigned)tn && bcur <= blast;
    int hin = dpp_from_prev(hcarry, 1);
    hin = top ? 1 : hin;             // the band's upper edge (row 0 of the matrix while the window has not moved: the same +1)
    uint32_t nP[MB_NW], nM[MB_NW];
#pragma unroll
    for (int w = 0; w < MB_NW; ++w) {
      uint32_t E = Eq[w];
      const uint32_t hinNeg = (hin < 0) ? 1u : 0u;   // edlib.cpp:390-470 (Hyyro's block step), 32-bit words
      const uint32_t Xv = E | Mv[w];
      E |= hinNeg;
      const uint32_t Xh = (((E & Pv[w]) + Pv[w]) ^ Pv[w]) | E;
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
    for (int w = 0; w < MB_NW; ++w) {
      Pv[w] = valid ? nP[w] : Pv[w];
      Mv[w] = valid ? nM[w] : Mv[w];
    }
    hcarry = valid ? hin : hcarry;
    score += valid ? hin : 0;
    // ---- the coming iteration: a slide (scalar schedule) moves every block of a pair that still has blocks below its window up
    //      one lane -- bit-vectors, score and pending delta with it; the bottom lane starts the fresh block below the one it held:
    //      all vertical deltas +1 from that block's last row in the column before (the band's lower edge)
    countdown -= 1;
    bool slid = false;
    if (countdown == 0) {
      countdown = MB_BH + 1;
      slid = slides_left > 0;
      uint32_t nPv[MB_NW], nMv[MB_NW];
#pragma unroll
      for (int w = 0; w < MB_NW; ++w) {
        nPv[w] = (uint32_t)dpp_from_next((int)Pv[w], -1);
        nMv[w] = (uint32_t)dpp_from_next((int)Mv[w], 0);
      }
      const int nhc = dpp_from_next(hcarry, 1), nsc = dpp_from_next(score, 0);
      const int fresh = score - hcarry + MB_BH;
#pragma unroll
      for (int w = 0; w < MB_NW; ++w) {
        Pv[w] = slid ? (bottom ? 0xffffffffu : nPv[w]) : Pv[w];
        Mv[w] = slid ? (bottom ? 0u : nMv[w]) : Mv[w];
      }
      hcarry = slid ? (bottom ? 1 : nhc) : hcarry;
      score = slid ? (bottom ? fresh : nsc) : score;
      bcur += slid ? 1 : 0;
      slides_left -= slid ? 1 : 0;
    }
    {
      const int codeNext = slid ? codeC : codeN;      // (a slid block stays in its column)
      c += slid ? 0 : 1;
      codeC = codeNext;
      codeN = load_code(c + 1);
      const uint32_t* eqp = &M.eq[eqoff + (min(bcur, max(blast, 0)) * 4 + codeNext) * MB_NW];   // (an iteration that is not valid discards what it computes)
#pragma unroll
      for (int w = 0; w < MB_NW; ++w) Eq[w] = eqp[w];
    }
  }
  // the lane that holds the pair's last block: D[pn][tn] = its bottom score minus the vertical deltas below row pn
  if (run && bcur == blast) {
    int sc = score;
#pragma unroll
    for (int w = 0; w < MB_NW; ++w) {
      const int lo = blast * MB_BH + w * 32;
      const int nbits = min(32, max(0, lo + 32 - pn));
      if (nbits > 0) {
        const uint32_t mk = (nbits >= 32) ? 0xffffffffu : (~0u << (32 - nbits));
        sc -= __popc(Pv[w] & mk);
        sc += __popc(Mv[w] & mk);
      }
    }
    M.res[g] = (sc <= k) ? sc : -1;
  }
  DH_SYNC();
}

}  // namespace dh
