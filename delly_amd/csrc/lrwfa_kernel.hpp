// lrwfa_kernel.hpp -- gfx950 device code for msaWfa (src/assemble.h:547-726), the long-read
// consensus of insertion junctions:
//   7-mer diagonal seeding (fillKmerTable :501-520, bestDiagonal :522-545) + trimmed NW distances
//   medoid order, 80 % cut                                    (:576-596)
//   superstring of the selected reads (NW PATH + buildSuperstring :90-133)   (:598-660)
//   progressive HW PATH alignment with the extended-IUPAC equalities, consensusWfa (:262-336),
//   convertAlignment HW (:24-88)                                         (:662-686)
//   consensus (src/msa.h:111-173), _trimConsensus against the reference anchors (:338-365)
// One junction per wavefront, on the edlib-equivalent strip machinery of lrmsa_kernel.hpp.
#pragma once
#include "lrins_kernel.hpp"

namespace dh {

constexpr int WFA_KMER = 7;                 // DELLY_KMER, src/tags.h:19
constexpr int WFA_KTAB = 65536;             // std::pow(4, DELLY_KMER + 1)
constexpr uint32_t WFA_DUP = 0xffffffffu;   // DELLY_DUPLICATE, src/tags.h:15
constexpr int WFA_ACAP_MAX = LM_RMASK - 1;  // alignment columns / superstring capacity (row keys hold LM_RBITS bits); per batch: LrWfaArgs::acap
constexpr int WFA_PCAP = 4096;              // reference anchor (prefix / suffix) capacity

struct LrWfaArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  const uint8_t* const* chr_seq;
  const int64_t* chr_len;
  dellyhip_params p;
  dellyhip_result* res;
  uint8_t* out_blob;
  uint64_t out_stride;
  int32_t out_cons_cap;
  int32_t* cons_len;
  const int32_t* work_list;  // junction indices (nullptr: 0..n_work-1)
  int32_t n_work;
  int32_t use_anchors;       // 1: prefix / suffix from the chromosome (src/assemble.h:855-856); 0: given / none
  const uint8_t* prefix;     // single-call mode
  const uint8_t* suffix;
  int32_t prefix_len, suffix_len;
  uint8_t* ws;
  uint64_t ws_stride;
  int32_t ncap;              // read length capacity
  int32_t acap;              // alignment columns / superstring capacity of this batch (<= WFA_ACAP_MAX)
  uint64_t off_alnB, off_astr, off_bnd, off_ops, off_tmp, off_cons, off_dirs, off_tabI, off_tabJ, off_diag, off_supA, off_supB,
      off_pre, off_suf, off_edit;   // alnA at 0
  uint64_t strip_words;
  const int32_t* edit_all;   // batch mode: the pairwise scores of every junction (LM_NR x LM_NR each), filled by wfa_pairs_kernel
};

__device__ __forceinline__ uint32_t wfa_char_to_int(uint8_t c) {   // charToInt, assemble.h:475-498
  // A, B -> 0; C, D -> 1; G, E -> 2; T, F -> 3; anything else 0: a 2-bit table over 'A' .. 'A' + 31 in one 64-bit constant (the
  // chain of comparisons compiles to divergent branches, eleven letters per k-mer and a k-mer per read position)
  const uint32_t idx = (uint32_t)c - (uint32_t)'A';
  return (idx < 32u) ? (uint32_t)((0xc000002e50ull >> (2u * idx)) & 3ull) : 0u;
}
__device__ __forceinline__ uint32_t wfa_hash(const uint8_t* s, int p) {
  // the seven letters in ONE eight-byte load (seven byte loads per position were seven round trips in flight per lane; the blob and
  // the workspace strings are padded, the eighth byte is ignored)
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef u32x2 __attribute__((aligned(1))) u32x2_u;
  const u32x2 v = *reinterpret_cast<const u32x2_u*>(s + p);
  uint32_t h = 0;
#pragma unroll
  for (int t = 0; t < WFA_KMER; ++t) h = h * 4u + wfa_char_to_int((uint8_t)((v[t >> 2] >> ((t & 3) * 8)) & 0xff));
  return h;
}

// fillKmerTable (len >= 7): tab[h] = 1-based start of the only occurrence, WFA_DUP when repeated, 0 when absent.
// The table must be all zero on entry; wfa_clear_table undoes exactly what this call touched.
// (four positions per lane in flight: the compare-and-swap returns the old entry, a round trip per position when done one by one)
__device__ __forceinline__ void wfa_fill_table(const uint8_t* s, int len, uint32_t* tab, int lane) {
  const int last = len - WFA_KMER;
  for (int p0 = lane; p0 <= last; p0 += 4 * WAVE) {
    uint32_t h[4], old[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) h[u] = (p0 + u * WAVE <= last) ? wfa_hash(s, p0 + u * WAVE) : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) old[u] = (p0 + u * WAVE <= last) ? atomicCAS(&tab[h[u]], 0u, (uint32_t)(p0 + u * WAVE + 1)) : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u * WAVE <= last && old[u] != 0u) atomicMax(&tab[h[u]], WFA_DUP);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ void wfa_clear_table(const uint8_t* s, int len, uint32_t* tab, int lane) {
  for (int p = lane; p <= len - WFA_KMER; p += WAVE) tab[wfa_hash(s, p)] = 0u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// bestDiagonal (assemble.h:522-545).  tabI / tabJ filled for sI / sJ; diag: scratch of lenI + lenJ + 64 words.
__device__ __forceinline__ int wfa_best_diagonal(const uint8_t* sJ, int lenI, int lenJ, const uint32_t* tabI,
                                                 const uint32_t* tabJ, uint32_t* diag, int lane) {
  const int dn = lenI + lenJ;
  for (int d = lane; d < dn + 64; d += WAVE) diag[d] = 0u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // every k-mer that is unique in both reads votes once for its diagonal
  for (int p0 = lane; p0 <= lenJ - WFA_KMER; p0 += 4 * WAVE) {   // (four positions per lane in flight)
    uint32_t hj[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = p0 + u * WAVE <= lenJ - WFA_KMER;
      const uint32_t h = in ? wfa_hash(sJ, p0 + u * WAVE) : 0u;
      hj[u] = in ? tabJ[h] : 0u;
      hi[u] = in ? tabI[h] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u * WAVE <= lenJ - WFA_KMER && hj[u] == (uint32_t)(p0 + u * WAVE + 1) && hi[u] != 0u && hi[u] != WFA_DUP)
        atomicAdd(&diag[lenJ + (int)hi[u] - (int)hj[u]], 1u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // window of 20 diagonals (:533-544): W(19) = sum diag[0..19] -> bestDiag 10; for d >= 20
  // W(d) = sum diag[d-19..d] -> bestDiag d - 10, taken only when strictly larger: the first maximum wins
  const int window = 20;
  unsigned long long best = 0;   // (W + 1) << 20 | (0xfffff - d)
  for (int d = window - 1 + lane; d < dn; d += WAVE) {
    uint32_t W = 0;
#pragma unroll
    for (int x = 0; x < window; ++x) W += diag[d - x];
    const unsigned long long key = ((unsigned long long)(W + 1u) << 20) | (unsigned long long)(0xfffff - d);
    best = key > best ? key : best;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int lo = __shfl_xor((int)(best & 0xffffffffull), o), hi = __shfl_xor((int)(best >> 32), o);
    const unsigned long long w = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
    best = w > best ? w : best;
  }
  int bestDiag = window / 2;   // also the answer when there are fewer than 20 diagonals
  if (best != 0ull) {
    const int dstar = 0xfffff - (int)(best & 0xfffffull);
    bestDiag = (dstar == window - 1) ? window / 2 : dstar - window / 2;
  }
  return rfl(bestDiag) - lenJ;
}

// ---- the pairwise stage of msaWfa for a whole batch: one (junction, read pair) per wavefront ---------------------------
struct WfaPairArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  const int32_t* pair_first;   // first item of junction j (prefix sums of n (n - 1) / 2 over the svt 4 junctions), n_junc + 1 entries
  int32_t n_junc, n_items;
  int32_t ncap, acap;          // the limits lrwfa_junction applies to every read of a junction
  int32_t* edit;               // edit[j * LM_NR * LM_NR + a * LM_NR + b]
  uint8_t* ws;                 // per block: k-mer tables of both reads (all zero between items), diagonal votes, strip buffers
  uint64_t ws_stride, off_tabJ, off_diag, off_hb, hb_half;
  uint32_t* next;              // work counter
  int32_t band_g, band_k, band_wl;   // banded distances (myers_band.hpp): pairs per wavefront and pass (0: off), band, lanes per pair
  const int32_t* seeds;        // per item (oI, oJ, lI, lJ) from wfa_seed_kernel (nullptr: seed here); oI = WFS_UNSEEDED: seed here
};
constexpr int WFS_UNSEEDED = -2;

// ---- the diagonal seeding of msaWfa's pairwise stage with its tables in LDS (round 6) ----------------------------------------
// wfa_pairs_kernel's own seeding keeps two 64 Ki-entry tables per wavefront in HBM and fills / probes / clears them with random
// four-byte accesses: 29 ms per 512 junctions, bound by HBM's random-access rate (2 GB of tables for 4 096 wavefronts).  But a
// 7-mer hash is below 4^7 = 16 384, and positions fit 16 bits: ONE table of 16 384 words holds read I's entry in the low half
// and read J's in the high half (0 absent, 0xFFFF repeated, else the 1-based start), 64 KB of LDS; the diagonal histogram
// (votes <= read length) is 16-bit counters, two per word.  One wavefront takes a ROW of a junction's pair matrix -- read a
// against every b > a -- so read a's entries are written once per row, not once per pair.  Results: the trimmed strings of
// every pair, (oI, oJ, lI, lJ), for the distance kernel.  Reads whose histogram does not fit are left to the old path.
constexpr int WFS_TAB = 16384;              // 4^DELLY_KMER
constexpr int WFS_DIAG = 8192;              // histogram entries (lenI + lenJ + 64 must fit; 64 + 16 KB of LDS: two wavefronts per CU)
struct WfaSeedArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  const int32_t* pair_first;   // as WfaPairArgs
  const int32_t* list;         // the insertion junctions
  int32_t n_list, max_rows;    // rows per junction at most (longest read list - 1)
  int32_t ncap, acap;
  int32_t* seeds;              // 4 ints per item
  uint32_t* next;              // work counter
};

// f(h, p) for every 7-mer start p of s[0 .. len) with len <= 4096 + 6: lane L takes the 64 consecutive starts [64 L, 64 L + 64) -- its
// 80 bytes arrive in five 16-byte loads that are all in flight together (a load per position was a round trip per loop iteration),
// and the hash rolls from one start to the next (two bits in, two bits out) instead of being rebuilt from seven letters
template <typename F>
__device__ __forceinline__ void wfs_for_each_kmer(const uint8_t* s_, int len, int lane, F f) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_u;
  const gptr_cu8 s = (gptr_cu8)s_;
  const int p0 = lane * 64;
  const int last = len - WFA_KMER;
  if (p0 > last) return;
  u32x4 v[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_u*>(s + p0 + 16 * q);   // (padded strings)
  auto code = [&](int q) -> uint32_t { return wfa_char_to_int((uint8_t)((v[q >> 4][(q >> 2) & 3] >> ((q & 3) * 8)) & 0xff)); };
  uint32_t h = 0;
#pragma unroll
  for (int t = 0; t < WFA_KMER - 1; ++t) h = h * 4u + code(t);
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    h = (h * 4u + code(q + WFA_KMER - 1)) & (uint32_t)(WFS_TAB - 1);
    if (p0 + q <= last) f(h, p0 + q);
  }
}

__global__ __launch_bounds__(WAVE) void wfa_seed_kernel(WfaSeedArgs A) {
  __shared__ uint32_t tab[WFS_TAB];
  __shared__ uint32_t diag[WFS_DIAG / 2];
  const int lane = threadIdx.x;
  for (int q = lane; q < WFS_TAB; q += WAVE) tab[q] = 0;
  __syncthreads();
  const int n_rows = __builtin_amdgcn_readfirstlane(A.n_list * A.max_rows);
  auto fetch = [&]() -> int {
    int v = 0;
    if (lane == 0) v = (int)atomicAdd(A.next, 1u);
    return __builtin_amdgcn_readfirstlane(v);
  };
  auto cnt = [&](int idx) -> uint32_t { return (diag[idx >> 1] >> (16 * (idx & 1))) & 0xffffu; };
  for (int row = fetch(); row < n_rows; row = fetch()) {
    // rows in a-major order: the long rows (a = 0: every other read of the junction) are handed out first
    const int a = row / A.n_list, j = A.list[row - a * A.n_list];
    const dellyhip_junction J = A.junc[j];
    const int N = J.n_seq;
    if (N < 2 || N > LM_NR || a > N - 2) continue;
    const uint64_t oa = A.seq_off[J.seq_first + a];
    const int lenI = (int)(A.seq_off[J.seq_first + a + 1] - oa);
    const uint8_t* sI = A.seq_blob + oa;
    const bool okI = !(lenI > A.ncap || lenI > A.acap - 2 || lenI < WFA_KMER + 1);
    int32_t* out = A.seeds + 4 * ((size_t)A.pair_first[j] + (size_t)a * (N - 1) - (size_t)a * (a - 1) / 2);   // the row's first item
    const bool rowfits = lenI <= 4096;
    if (okI && rowfits) {   // fillKmerTable, low halves (the high halves are zero between pairs)
      wfs_for_each_kmer(sI, lenI, lane, [&](uint32_t h, int p) {
        const uint32_t old = atomicCAS(&tab[h], 0u, (uint32_t)(p + 1));
        if (old != 0u) atomicOr(&tab[h], 0xffffu);
      });
    }
    __syncthreads();
    for (int b = a + 1; b < N; ++b, out += 4) {
      const uint64_t ob = A.seq_off[J.seq_first + b];
      const int lenJ = (int)(A.seq_off[J.seq_first + b + 1] - ob);
      const uint8_t* sJ = A.seq_blob + ob;
      const bool okJ = !(lenJ > A.ncap || lenJ > A.acap - 2 || lenJ < WFA_KMER + 1);
      if (!okI || !okJ) {   // (wfa_pairs_kernel skips the pair on the same test)
        if (lane == 0) { out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0; }
        continue;
      }
      const int dn = lenI + lenJ;
      if (dn + 64 > WFS_DIAG || !rowfits || lenJ > 4096) {   // the histogram / a lane's stretch does not fit: seeded by wfa_pairs_kernel
        if (lane == 0) out[0] = WFS_UNSEEDED;
        continue;
      }
      wfs_for_each_kmer(sJ, lenJ, lane, [&](uint32_t h, int p) {   // read J's entries, high halves
        const uint32_t old = atomicOr(&tab[h], (uint32_t)(p + 1) << 16);
        if ((old >> 16) != 0u) atomicOr(&tab[h], 0xffff0000u);
      });
      for (int d = lane; d < (dn + 64 + 1) / 2; d += WAVE) diag[d] = 0u;
      __syncthreads();
      // bestDiagonal (assemble.h:522-545): every k-mer that is unique in both reads votes once for its diagonal
      wfs_for_each_kmer(sJ, lenJ, lane, [&](uint32_t h, int p) {
        const uint32_t w = tab[h];
        const uint32_t hj = w >> 16, hi = w & 0xffffu;
        if (hj == (uint32_t)(p + 1) && hi != 0u && hi != 0xffffu) {
          const int idx = lenJ + (int)hi - (int)hj;
          atomicAdd(&diag[idx >> 1], 1u << (16 * (idx & 1)));
        }
      });
      __syncthreads();
      // window of 20 diagonals (:533-544): the first maximum wins (see wfa_best_diagonal)
      const int window = 20;
      unsigned long long best = 0;   // (W + 1) << 20 | (0xfffff - d)
      {   // every lane slides the window over a contiguous stretch of diagonals: two reads per diagonal instead of twenty
        const int nd = dn - (window - 1);                       // windows end at d = window - 1 .. dn - 1
        const int per = (max(nd, 0) + WAVE - 1) / WAVE;
        const int d0 = window - 1 + lane * per, d1 = min(dn, d0 + per);
        if (d0 < d1) {
          uint32_t W = 0;
#pragma unroll
          for (int x = 0; x < window; ++x) W += cnt(d0 - x);
          for (int d = d0;; ) {
            const unsigned long long key = ((unsigned long long)(W + 1u) << 20) | (unsigned long long)(0xfffff - d);
            best = key > best ? key : best;
            if (++d >= d1) break;
            W += cnt(d) - cnt(d - window);
          }
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const int lo = __shfl_xor((int)(best & 0xffffffffull), o), hi = __shfl_xor((int)(best >> 32), o);
        const unsigned long long w = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
        best = w > best ? w : best;
      }
      int bestDiag = window / 2;   // also the answer when there are fewer than 20 diagonals
      if (best != 0ull) {
        const int dstar = 0xfffff - (int)(best & 0xfffffull);
        bestDiag = (dstar == window - 1) ? window / 2 : dstar - window / 2;
      }
      const int bd = rfl(bestDiag) - lenJ;
      wfs_for_each_kmer(sJ, lenJ, lane, [&](uint32_t h, int) { atomicAnd(&tab[h], 0xffffu); });   // read J leaves the table
      uint32_t oI, oJ, seqlen;
      if (bd >= 0) { seqlen = min((uint32_t)lenI - (uint32_t)bd, (uint32_t)lenJ); oI = (uint32_t)bd; oJ = 0; }
      else { seqlen = min((uint32_t)lenJ + (uint32_t)bd, (uint32_t)lenI); oI = 0; oJ = (uint32_t)(-bd); }
      const uint32_t lI = min(seqlen, (uint32_t)lenI - oI), lJ = min(seqlen, (uint32_t)lenJ - oJ);   // substr clamps
      if (lane == 0) { out[0] = (int32_t)oI; out[1] = (int32_t)oJ; out[2] = (int32_t)lI; out[3] = (int32_t)lJ; }
      __syncthreads();
    }
    if (okI && rowfits) wfs_for_each_kmer(sI, lenI, lane, [&](uint32_t h, int) { tab[h] = 0u; });
    __syncthreads();
  }
}

__global__ __launch_bounds__(WAVE) void wfa_pairs_kernel(WfaPairArgs A) {
  MyersBandLds& LB = myers_band_lds();
  MyersLds<MYERS_NW>& L = LB.full();
  const int lane = threadIdx.x;
  myers_lut_init(L.lut, lane);
  __syncthreads();
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  uint32_t* tabI = reinterpret_cast<uint32_t*>(ws);
  uint32_t* tabJ = reinterpret_cast<uint32_t*>(ws + A.off_tabJ);
  uint32_t* diag = reinterpret_cast<uint32_t*>(ws + A.off_diag);
  int8_t* hb = reinterpret_cast<int8_t*>(ws + A.off_hb);
  const int n_items = __builtin_amdgcn_readfirstlane(A.n_items);
  auto fetch = [&]() -> int {
    int v = 0;
    if (lane == 0) v = (int)atomicAdd(A.next, 1u);
    return __builtin_amdgcn_readfirstlane(v);
  };
  // one item: (junction, read pair) -> diagonal seeding -> the trimmed strings whose NW distance is wanted
  struct Seeded { int j, a, b; const uint8_t* sI; const uint8_t* sJ; uint32_t lI, lJ; bool skip; };
  auto seed = [&](int item) -> Seeded {
    Seeded S;
    int lo = 0, hi = A.n_junc;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (A.pair_first[mid] <= item) lo = mid;
      else hi = mid;
    }
    S.j = lo;
    const dellyhip_junction J = A.junc[S.j];
    const int N = J.n_seq;
    int rem = item - A.pair_first[S.j], a = 0;
    while (rem >= N - 1 - a) {
      rem -= N - 1 - a;
      ++a;
    }
    S.a = a;
    S.b = a + 1 + rem;
    const uint64_t oa = A.seq_off[J.seq_first + S.a], ob = A.seq_off[J.seq_first + S.b];
    const int lenI = (int)(A.seq_off[J.seq_first + S.a + 1] - oa), lenJ = (int)(A.seq_off[J.seq_first + S.b + 1] - ob);
    S.sI = S.sJ = nullptr;
    S.lI = S.lJ = 0;
    // (a junction with a read outside these limits is flagged by lrwfa_junction and never looks at its scores)
    S.skip = lenI > A.ncap || lenI > A.acap - 2 || lenI < WFA_KMER + 1 || lenJ > A.ncap || lenJ > A.acap - 2 || lenJ < WFA_KMER + 1;
    if (S.skip) return S;
    const uint8_t* sI = A.seq_blob + oa;
    const uint8_t* sJ = A.seq_blob + ob;
    if (A.seeds) {   // wfa_seed_kernel has done the diagonal seeding with its tables in LDS
      const int32_t* sd4 = A.seeds + 4 * (size_t)item;
      const int s0 = rfl(sd4[0]);
      if (s0 != WFS_UNSEEDED) {
        S.lI = (uint32_t)rfl(sd4[2]);
        S.lJ = (uint32_t)rfl(sd4[3]);
        S.sI = sI + s0;
        S.sJ = sJ + rfl(sd4[1]);
        return S;
      }
    }
    wfa_fill_table(sI, lenI, tabI, lane);
    wfa_fill_table(sJ, lenJ, tabJ, lane);
    const int bd = wfa_best_diagonal(sJ, lenI, lenJ, tabI, tabJ, diag, lane);
    wfa_clear_table(sJ, lenJ, tabJ, lane);
    wfa_clear_table(sI, lenI, tabI, lane);
    uint32_t oI, oJ, seqlen;
    if (bd >= 0) { seqlen = min((uint32_t)lenI - (uint32_t)bd, (uint32_t)lenJ); oI = (uint32_t)bd; oJ = 0; }
    else { seqlen = min((uint32_t)lenJ + (uint32_t)bd, (uint32_t)lenI); oI = 0; oJ = (uint32_t)(-bd); }
    S.lI = min(seqlen, (uint32_t)lenI - oI);   // substr clamps
    S.lJ = min(seqlen, (uint32_t)lenJ - oJ);
    S.sI = sI + oI;
    S.sJ = sJ + oJ;
    return S;
  };
  auto full_distance = [&](const Seeded& S) -> int {
    const uint32_t lI = S.lI, lJ = S.lJ;
    if (lI == 0 || lJ == 0) return (int)max(lI, lJ);
    if (lI > (uint32_t)MYERS_ROWS && lJ > (uint32_t)MYERS_ROWS)
      return (A.hb_half && (uint64_t)max(lI, lJ) + 16 <= A.hb_half) ? rfl(myers_nw_big(S.sI, (int)lI, S.sJ, (int)lJ, hb, hb + A.hb_half, lane)) : -1;
    if (lI <= lJ) return rfl(myers_nw_auto(L, S.sI, (int)lI, S.sJ, (int)lJ, lane));   // (pattern = the shorter string: the distance is symmetric)
    return rfl(myers_nw_auto(L, S.sJ, (int)lJ, S.sI, (int)lI, lane));
  };
  auto finish = [&](const Seeded& S, int d) {
    const int score = (d < 0) ? -1 : (d * 1000) / (int)max(S.lI, S.lJ);
    if (lane == 0) {
      int32_t* E = A.edit + (size_t)S.j * LM_NR * LM_NR;
      E[S.a * LM_NR + S.b] = score;
      E[S.b * LM_NR + S.a] = score;
    }
  };
  if (A.band_g >= 2) {
    // band_g items at a time: their seedings one after the other (the k-mer tables are the wavefront's), then ONE banded pass
    // over all their trimmed strings (myers_band.hpp); what the band does not certify gets the full pass
    __shared__ Seeded sd[MB_G];
    const int G = A.band_g;
    auto fetch_group = [&]() -> int {
      int v = 0;
      if (lane == 0) v = (int)atomicAdd(A.next, (uint32_t)G);
      return __builtin_amdgcn_readfirstlane(v);
    };
    for (int first = fetch_group(); first < n_items; first = fetch_group()) {
      const int cnt = min(G, n_items - first);
      MbLds& M = LB.band();
      for (int q = 0; q < cnt; ++q) {
        const Seeded S = seed(first + q);
        if (lane == 0) {
          sd[q] = S;
          const bool ij = S.lI <= S.lJ;
          M.item[q].pat = ij ? S.sI : S.sJ;
          M.item[q].txt = ij ? S.sJ : S.sI;
          M.item[q].pn = S.skip ? 0 : (int)min(S.lI, S.lJ);     // (0 rows: not a banded item)
          M.item[q].tn = S.skip ? 0 : (int)max(S.lI, S.lJ);
        }
      }
      __syncthreads();
      myers_band_multi(cnt, A.band_k, A.band_wl, lane);
      int res[MB_G];
#pragma unroll
      for (int q = 0; q < MB_G; ++q) res[q] = (q < cnt) ? M.res[q] : 0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < MB_G; ++q) {
        if (q < cnt) {
          const Seeded S = sd[q];
          if (!S.skip) finish(S, (rfl(res[q]) >= 0) ? rfl(res[q]) : full_distance(S));   // (the full pass rewrites the tables' bytes: the results are in registers)
        }
      }
      __syncthreads();
    }
    return;
  }
  for (int item = fetch(); item < n_items; item = fetch()) {
    const Seeded S = seed(item);
    if (S.skip) continue;
    finish(S, full_distance(S));
  }
}

#ifdef DH_LR_TIMING
#define LRW_LIMIT(site) ((lane == 0 ? printf("lrwfa junction %d: limit at line %d\n", j, site) : 0), DELLYHIP_E_LIMIT)
#else
#define LRW_LIMIT(site) DELLYHIP_E_LIMIT
#endif
// msaWfa for one junction
__device__ void lrwfa_junction(const LrWfaArgs& A, int j, LrMsaLds& L, uint8_t* ws, int lane) {
  const dellyhip_junction J = A.junc[j];
  dellyhip_result* out = &A.res[j];
  uint8_t* cons_out = A.out_blob + (size_t)j * A.out_stride;
  const int N = J.n_seq;
  int status = 0, cons_len = 0, rows = 0;
#ifdef DH_LR_TIMING
  unsigned long long tw0 = 0, tw1 = 0, tw2 = 0;
#endif
  uint8_t* alnA = ws;
  uint8_t* alnB = ws + A.off_alnB;
  uint8_t* astr = ws + A.off_astr;
  int32_t* bnd = reinterpret_cast<int32_t*>(ws + A.off_bnd);
  uint8_t* ops = ws + A.off_ops;
  uint8_t* tmp = ws + A.off_tmp;
  uint8_t* cbuf = ws + A.off_cons;
  uint32_t* dirs = reinterpret_cast<uint32_t*>(ws + A.off_dirs);
  uint32_t* tabI = reinterpret_cast<uint32_t*>(ws + A.off_tabI);
  uint32_t* tabJ = reinterpret_cast<uint32_t*>(ws + A.off_tabJ);
  uint32_t* diag = reinterpret_cast<uint32_t*>(ws + A.off_diag);
  uint8_t* supA = ws + A.off_supA;
  uint8_t* supB = ws + A.off_supB;
  uint8_t* pre = ws + A.off_pre;
  uint8_t* suf = ws + A.off_suf;
  int32_t* E = A.edit_all ? const_cast<int32_t*>(A.edit_all) + (size_t)j * LM_NR * LM_NR : reinterpret_cast<int32_t*>(ws + A.off_edit);
  const int bnd_stride = max(A.ncap, A.acap) + 128;
  const int acap = A.acap;
  const int ops_cap = 2 * max(acap, A.ncap) + 32;
  const uint8_t* blob = A.seq_blob;
  if (N >= 1) {
    if (N > LM_NR) status = LRW_LIMIT(476);
    if (!status) {
      for (int r = lane; r < N; r += WAVE) {
        const uint64_t a = A.seq_off[J.seq_first + r], b = A.seq_off[J.seq_first + r + 1];
        L.roff[r] = a;
        L.rlen[r] = (int32_t)(b - a);
      }
      __syncthreads();
      for (int r = 0; r < N; ++r)
        if (L.rlen[r] > A.ncap || L.rlen[r] > acap - 2 || L.rlen[r] < WFA_KMER + 1) status = LRW_LIMIT(485);
    }
    // reads made of A, C, G, T only (the usual case): their plain-equality alignments may use the compare-free
    // bit-vector passes (lm_pure_acgt); the superstring is pure while every read merged into it is
    unsigned long long pure_reads = 0;
    if (!status)
      for (int r = 0; r < N; ++r)
        if (lm_pure_acgt(blob + L.roff[r], 1, L.rlen[r], lane)) pure_reads |= 1ull << r;
    // reference anchors (src/assemble.h:855-856)
    int pn = 0, sn = 0;
    if (!status) {
      if (A.use_anchors) {
        const uint8_t* seq = A.chr_seq[J.chr];
        const int seqlen = (int)(uint32_t)A.chr_len[J.chr];
        const int p0 = max(J.sv_start - A.p.min_cons_window, 0), p1 = J.sv_start;
        const int s0 = J.sv_start, s1 = min(seqlen, J.sv_start + A.p.min_cons_window);
        pn = max(0, p1 - p0);
        sn = max(0, s1 - s0);
        if (pn > WFA_PCAP || sn > WFA_PCAP) status = LRW_LIMIT(503);
        else {
          for (int k = lane; k < pn; k += WAVE) pre[k] = upc(seq[p0 + k]);
          for (int k = lane; k < sn; k += WAVE) suf[k] = upc(seq[s0 + k]);
        }
      } else {
        pn = A.prefix_len;
        sn = A.suffix_len;
        if (pn > WFA_PCAP || sn > WFA_PCAP) status = LRW_LIMIT(511);
        else {
          for (int k = lane; k < pn; k += WAVE) pre[k] = A.prefix[k];
          for (int k = lane; k < sn; k += WAVE) suf[k] = A.suffix[k];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
#ifdef DH_LR_TIMING
    tw0 = tw1 = tw2 = wall_clock64();
    LRT_START();
#endif
    if (!status) {
      // ---- pairwise scores: diagonal seeding, trimmed NW distance, per mille of the length (:551-574)
      // (batch mode: wfa_pairs_kernel has filled E -- N (N - 1) / 2 independent pairs are throughput work for the whole
      //  chip, not 2/3 of this wavefront's latency)
      for (int a = 0; a < N && !A.edit_all; ++a) {
        const uint8_t* sI = blob + L.roff[a];
        const int lenI = L.rlen[a];
        wfa_fill_table(sI, lenI, tabI, lane);
        for (int b = a + 1; b < N; ++b) {
          const uint8_t* sJ = blob + L.roff[b];
          const int lenJ = L.rlen[b];
          wfa_fill_table(sJ, lenJ, tabJ, lane);
          const int bd = wfa_best_diagonal(sJ, lenI, lenJ, tabI, tabJ, diag, lane);
          wfa_clear_table(sJ, lenJ, tabJ, lane);
          uint32_t oI, oJ, seqlen;
          if (bd >= 0) { seqlen = min((uint32_t)lenI - (uint32_t)bd, (uint32_t)lenJ); oI = (uint32_t)bd; oJ = 0; }
          else { seqlen = min((uint32_t)lenJ + (uint32_t)bd, (uint32_t)lenI); oI = 0; oJ = (uint32_t)(-bd); }
          uint32_t lI = min(seqlen, (uint32_t)lenI - oI), lJ = min(seqlen, (uint32_t)lenJ - oJ);   // substr clamps
          int d;
          if (lI == 0 || lJ == 0) d = (int)max(lI, lJ);
          else {   // (strings beyond the rows of one bit-vector pass: strips, boundary deltas parked in the bnd area)
            int8_t* hb = reinterpret_cast<int8_t*>(bnd);
            d = rfl(myers_nw_big(sI + oI, (int)lI, sJ + oJ, (int)lJ, hb, hb + 2 * bnd_stride, lane));
          }
          const int score = (d * 1000) / (int)max(lI, lJ);
          if (lane == 0) {
            E[a * LM_NR + b] = score;
            E[b * LM_NR + a] = score;
          }
        }
        wfa_clear_table(sI, lenI, tabI, lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#ifdef DH_LR_TIMING
      tw1 = wall_clock64();
#endif
      // ---- medoid, order, 80 % cut (:576-596) -- as in msaEdlib
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
      uint32_t lastIdx = (uint32_t)(0.8 * N);
      if (lastIdx < 3) lastIdx = 3;
      const int nsel = min((int)lastIdx, N);
      if (lane < N) {
        const int kx = (lane == bestIdx) ? 0 : E[bestIdx * LM_NR + lane];
        int rank = 0;
        for (int y = 0; y < N; ++y) {
          const int ky = (y == bestIdx) ? 0 : E[bestIdx * LM_NR + y];
          rank += (ky < kx || (ky == kx && y < lane)) ? 1 : 0;
        }
        if (rank < LM_NR) L.sel[rank] = lane;
      }
      __syncthreads();
      // ---- superstring (:598-660)
      uint8_t* sup = supA;
      uint8_t* sup2 = supB;
      int sl = L.rlen[L.sel[0]];
      bool sup_pure = ((pure_reads >> L.sel[0]) & 1ull) != 0;
      {
        const uint8_t* r0 = blob + L.roff[L.sel[0]];
        for (int k = lane; k < sl; k += WAVE) sup[k] = r0[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      for (int step = 1; step < nsel && !status; ++step) {
        const uint8_t* rd = blob + L.roff[L.sel[step]];
        const uint32_t lenI = (uint32_t)sl, lenJ = (uint32_t)L.rlen[L.sel[step]];
        wfa_fill_table(sup, (int)lenI, tabI, lane);
        wfa_fill_table(rd, (int)lenJ, tabJ, lane);
        const int bd = wfa_best_diagonal(rd, (int)lenI, (int)lenJ, tabI, tabJ, diag, lane);
        wfa_clear_table(sup, (int)lenI, tabI, lane);
        wfa_clear_table(rd, (int)lenJ, tabJ, lane);
        LRT_LAP(1);
        uint32_t preI, postI, preJ, postJ, seqlen;
        if (bd >= 0) {
          seqlen = min(lenI - (uint32_t)bd, lenJ);
          preI = (uint32_t)bd; postI = lenI - ((uint32_t)bd + seqlen); preJ = 0; postJ = lenJ - seqlen;
        } else {
          seqlen = min(lenJ + (uint32_t)bd, lenI);
          preI = 0; postI = lenI - seqlen; preJ = (uint32_t)(-bd); postJ = lenJ - ((uint32_t)(-bd) + seqlen);
        }
        if (preI > preJ && postI > postJ) {
          // nested: the superstring already contains the read
        } else if (preJ > preI && postJ > postI) {
          for (int k = lane; k < (int)lenJ; k += WAVE) sup[k] = rd[k];
          sl = (int)lenJ;
          sup_pure = ((pure_reads >> L.sel[step]) & 1ull) != 0;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        } else {
          const uint8_t* sI = (bd >= 0) ? sup + bd : sup;
          const uint8_t* sJ = (bd >= 0) ? rd : rd + (-bd);
          if ((int)seqlen > acap - 2 || seqlen == 0) { status = LRW_LIMIT(631); break; }
          // edlibAlign(seqI, seqJ, NW, PATH): query = seqI (columns), target = seqJ (rows)
          const bool rd_pure = ((pure_reads >> L.sel[step]) & 1ull) != 0;
          const int pmode = (sup_pure && rd_pure) ? (LM_EQ | LM_EQFAST) : 0;   // (identity among ACGT: same op string)
#ifdef DH_LR_TIMING
          const unsigned long long lrt_p0 = wall_clock64();
#endif
          const int nops = lm_nw_path(sJ, (int)seqlen, sI, (int)seqlen, pmode, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
#ifdef DH_LR_TIMING
          LRT_ADD(2, wall_clock64() - lrt_p0);
          LRT_LAP(12);
#endif
          if (nops < 0) { status = LRW_LIMIT(643); break; }
          sup_pure = sup_pure && rd_pure;
          // buildSuperstring (:90-133)
          const bool f0 = preI > preJ;
          const int plen = f0 ? (int)preI : (int)preJ;
          for (int k = lane; k < plen; k += WAVE) sup2[k] = f0 ? sup[k] : rd[k];
          const int bp = nops / 2;
          int ib = (int)preI, jb = (int)preJ, ob = plen;
          for (int base = 0; base < nops; base += WAVE) {
            const int q = base + lane;
            const int op = (q < nops) ? (int)ops[q] : ED_MATCH;
            const bool first = (q < bp) ? f0 : !f0;
            const bool isI = (q < nops) && (op != ED_DELETE);   // consumes seqI
            const bool isJ = (q < nops) && (op != ED_INSERT);   // consumes seqJ
            const bool emit = (q < nops) && ((op == ED_DELETE) ? !first : (op == ED_INSERT) ? first : true);
            const unsigned long long mi = __ballot(isI), mj = __ballot(isJ), me = __ballot(emit);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int ii = ib + __popcll(mi & below), jj = jb + __popcll(mj & below), oo = ob + __popcll(me & below);
            if (emit) sup2[oo] = (op == ED_DELETE) ? rd[jj] : (op == ED_INSERT) ? sup[ii] : (first ? sup[ii] : rd[jj]);
            ib += __popcll(mi);
            jb += __popcll(mj);
            ob += __popcll(me);
          }
          const bool tailI = postI > postJ;
          const int tlen = tailI ? (int)postI : (int)postJ;
          if (ob + tlen > acap - 2) { status = LRW_LIMIT(668); break; }
          for (int k = lane; k < tlen; k += WAVE) sup2[ob + k] = tailI ? sup[ib + k] : rd[jb + k];
          sl = ob + tlen;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          LRT_LAP(3);
          uint8_t* sw = sup; sup = sup2; sup2 = sw;
        }
      }
#ifdef DH_LR_TIMING
      tw2 = wall_clock64();
#endif
      // ---- progressive HW alignment of every selected read (:662-686)
      uint8_t* cur = alnA;
      uint8_t* nxt = alnB;
      int arows = 1, acols = sl;
      if (!status) {
        for (int k = lane; k < acols; k += WAVE) cur[k] = sup[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      for (int step = 0; step < nsel && !status; ++step) {
        // consensusWfa (:262-336): only rows spanning the column vote
        for (int r = 0; r < arows; ++r) {
          int first = acols, last = 0;
          bool any = false;
          for (int base = 0; base < acols; base += WAVE) {
            const int c = base + lane;
            const unsigned long long bm = __ballot(c < acols && cur[(size_t)r * acap + c] != '-');
            if (bm) {
              if (!any) first = base + __builtin_ctzll(bm);
              last = base + 63 - __builtin_clzll(bm);
              any = true;
            }
          }
          if (lane == 0) { L.first[r] = first; L.last[r] = last; }   // readStart = cols, readEnd = 0 when the row is all gaps
        }
        __syncthreads();
        for (int col = lane; col < acols; col += WAVE) {
          // (the five counts as bytes of one word, the letter's slot from a nibble table: an if-else chain over per-lane letters is a tree
          //  of divergent branches; rows <= 255)
          unsigned long long packed = 0ull;
          for (int r = 0; r < arows; ++r) {
            const uint8_t ch = cur[(size_t)r * acap + col];
            const int v = letter_code_bf((uint8_t)(ch & 0xDF));   // A/a 0, C/c 1, G/g 2, T/t 3; N/n, '-' and everything else: slot 4
            const unsigned long long one = (col >= L.first[r] && col <= L.last[r]) ? 1ull : 0ull;
            packed += one << (8 * ((v < 0) ? 4 : v));
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
        const LmRes h = lm_hw(astr, acols, qy, qn, eqmode, true, true, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
#ifdef DH_LR_TIMING
        LRT_ADD(13, wall_clock64() - lrt_p0);
        LRT_LAP(12);
#endif
        if (h.nops < 0) { status = LRW_LIMIT(751); break; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // convertAlignment(query, align, HW, cigar) (:24-88)
        const int nops = h.nops;
        const int missingStart = h.startLoc;                       // tIdx = end - #nonINSERT = start - 1
        const int missingEnd = (h.endLoc < acols) ? acols - h.endLoc - 1 : 0;
        const int ncols = missingStart + nops + missingEnd;
        if (ncols > acap - 2) { status = LRW_LIMIT(759); break; }
        for (int c = lane; c < missingStart; c += WAVE) {
          for (int r = 0; r < arows; ++r) nxt[(size_t)r * acap + c] = cur[(size_t)r * acap + c];
          nxt[(size_t)arows * acap + c] = '-';
        }
        int tbase = (h.endLoc == -1) ? 0 : missingStart, qbase = 0;
        for (int base = 0; base < nops; base += WAVE) {
          const int jc = base + lane;
          const int op = (jc < nops) ? (int)ops[jc] : ED_MATCH;
          const unsigned long long mt = __ballot(jc < nops && op != ED_INSERT);
          const unsigned long long mq = __ballot(jc < nops && op != ED_DELETE);
          const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int ti = tbase + __popcll(mt & below), qi = qbase + __popcll(mq & below);
          if (jc < nops) {
            const int oc = missingStart + jc;
            for (int r = 0; r < arows; ++r) nxt[(size_t)r * acap + oc] = (op != ED_INSERT) ? cur[(size_t)r * acap + ti] : (uint8_t)'-';
            nxt[(size_t)arows * acap + oc] = (op != ED_DELETE) ? qy[qi] : (uint8_t)'-';
          }
          tbase += __popcll(mt);
          qbase += __popcll(mq);
        }
        for (int c = lane; c < missingEnd; c += WAVE) {
          const int oc = missingStart + nops + c;
          for (int r = 0; r < arows; ++r) nxt[(size_t)r * acap + oc] = cur[(size_t)r * acap + tbase + c];
          nxt[(size_t)arows * acap + oc] = '-';
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LRT_LAP(10);
        uint8_t* sw = cur; cur = nxt; nxt = sw;
        arows += 1;
        acols = ncols;
      }
      if (!status) {
        Node nd{cur, arows, acols, acap};
        int Lc = consensus_node(nd, A.p, cbuf, acap, L, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int o = 0;
        if (pn > 0 && sn > 0) {
          // _trimConsensus (:338-365)
          if (Lc < 1 || Lc > acap - 2) { Lc = max(Lc, 0); }
          if (Lc >= 1) {
            uint8_t* prev = sup2;   // reverse complement of the prefix (the superstring buffers are free by now)
            for (int k = lane; k < pn; k += WAVE) prev[k] = rc_at(pre, pn, k);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // (the consensus holds A, C, G, T only -- consensus_node; anchors without N etc. allow the bit-vector passes)
            const int amode = (lm_pure_acgt(pre, 1, pn, lane) && lm_pure_acgt(suf, 1, sn, lane) && lm_pure_acgt(cbuf, 1, Lc, lane)) ? (LM_EQ | LM_EQFAST) : 0;
            const LmRes f = lm_hw(cbuf, Lc, pre, pn, amode, false, false, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
            const LmRes r = lm_hw(cbuf, Lc, prev, pn, amode, false, false, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
            if (f.ed > r.ed) {   // reverseComplement(cs), util.h:549-563 semantics
              for (int k = lane; k < Lc; k += WAVE) astr[k] = rc_at(cbuf, Lc, k);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
              for (int k = lane; k < Lc; k += WAVE) cbuf[k] = astr[k];
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
            }
            const LmRes cp = lm_hw(cbuf, Lc, pre, pn, amode, true, false, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
            const LmRes cs2 = lm_hw(cbuf, Lc, suf, sn, amode, false, false, bnd, bnd_stride, dirs, A.strip_words, tmp, ops, ops_cap, lane);
            const uint32_t csStart = (uint32_t)cp.startLoc, csEnd = (uint32_t)cs2.endLoc;
            if (csStart < csEnd && csEnd < (uint32_t)Lc) { o = (int)csStart; Lc = (int)(csEnd - csStart); }
          }
        } else {
          int trim = (int)(0.05 * Lc);
          if (trim > 50) trim = 50;
          const int len = Lc - 2 * trim;
          if (len > 100) { o = trim; Lc = len; }
        }
        if (Lc > A.out_cons_cap) status = LRW_LIMIT(829);
        else {
          for (int k = lane; k < Lc; k += WAVE) cons_out[k] = cbuf[o + k];
          cons_len = Lc;
          rows = nsel;
        }
      }
    }
  }
#ifdef DH_LR_TIMING
  LRT_LAP(11);
  LRT_ADD(14, 1);
  LRT_ADD(15, wall_clock64() - tw0);
  (void)tw1; (void)tw2;
  if (lane == 0 && A.n_work <= 64) printf("lrwfa junction %d: N %d status %d cons_len %d rows %d\n", j, N, status, cons_len, rows);
#endif
  if (lane == 0) {
    out->sr_support = rows;
    out->status = status;
    A.cons_len[j] = status ? 0 : cons_len;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__global__ __launch_bounds__(WAVE) void lrwfa_kernel(LrWfaArgs A) {
  __shared__ LrMsaLds L;
  const int lane = threadIdx.x;
  uint8_t* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  for (int w = blockIdx.x; w < A.n_work; w += gridDim.x) {
    const int j = A.work_list ? A.work_list[w] : w;
    if (j < 0) continue;
    lrwfa_junction(A, j, L, ws, lane);
  }
}

}  // namespace dh
