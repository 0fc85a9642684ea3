// classify_kernel.hpp -- gfx950 device code: the split-read genotyping classifier, the worker body of
// process_batch (src/coverage.h:418-434): per AlignJob two _editDistanceHW calls (:107-115) -- edlib
// HW edit DISTANCE of a probe (26 .. ~60 bytes) inside a read (150 bytes) with threshold k -- and the
// 'R' / 'A' / 'N' decision.  The HW distance is unique, so only its value has to match edlib's.
//
// Shape of the work: millions of independent, tiny problems (a 64-row pattern against 150 columns),
// so ONE JOB PER LANE: each lane runs Myers' bit-vector recurrence (the arithmetic of
// src/edlib.cpp:390-470 calculateBlock) with the whole probe in one 64-bit word, both probes of
// the job in lock-step over the same read bytes.  The probes' equality masks sit in LDS, indexed
// [probe][word][letter][lane] (bank-conflict free, one ds_read_b64 per probe and column); the read is
// fetched 16 bytes per lane and load.  Probes of 65 .. 256 bytes (rare) go through the same code
// with four words per probe in a second, small launch; wavefronts whose probes all fit 32 rows use 32-bit words.
#pragma once
#include "myers_kernel.hpp"
#include "split_kernel.hpp"

namespace dh {

constexpr int CLS_MAXW = 4;                  // 64-bit words per probe in the wide launch
constexpr int CLS_PROBE_MAX = 64 * CLS_MAXW;
constexpr int CLS_NCODE = 6;                 // A C G T N + one all-zero slot (no row matches)
constexpr int CLS_PAD = 16;                  // bytes the device blob is padded with (16-byte read fetches)

struct ClsArgs {
  const dellyhip_align_job* jobs;
  const uint8_t* blob;
  dellyhip_align_result* res;
  uint64_t n_jobs;
  float flank_quality;
  int32_t* wide_list;    // one-word launch: jobs with a probe > 64 bytes are appended here ...
  int32_t* wide_count;   // ... and counted; the wide launch reads both
  int32_t* big_list;     // wide launch: jobs with a probe > 256 bytes -> classify_big_kernel (one job per wavefront)
  int32_t* big_count;
};

template <int NW>
struct ClsLds {
  uint64_t peq[2 * NW * CLS_NCODE * WAVE];   // [probe][word][code][lane]
  uint16_t lut[256];                         // byte -> code * WAVE (zero slot for everything outside ACGTN)
};

__device__ __forceinline__ int cls_code(int c) {
  const int v = ((unsigned)c > 255u) ? -1 : letter_code_bf((uint8_t)c);
  return v < 0 ? 5 : v;
}

// exact equality mask of probe rows [64w, 64w+64) for a byte outside ACGTN (only reached when a probe
// itself holds such a byte)
__device__ __noinline__ uint64_t cls_eq_slow(const uint8_t* probe, int m, int w, int c) {
  uint64_t Eq = 0;
  for (int q = 0; q < 64; ++q) {
    const int r = w * 64 + q;
    if (r < m && (int)probe[r] == c) Eq |= 1ull << q;
  }
  return Eq;
}

__device__ __forceinline__ int cls_wave_max(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
  return v;
}

__device__ __forceinline__ uint4 cls_load16(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// edlib's result for one probe: the HW distance, -1 beyond k  (src/edlib.cpp:157-170 empty operands,
// :564 k = min(queryLength, k), :650 colScore <= k)
__device__ __forceinline__ int cls_distance(int best, int m, int n, float fq) {
  if (m == 0 || n == 0) return m;
  const int k = (int)((2.0f * fq) * (float)m);   // edlibNewAlignConfig(2 * c.flankQuality * query.size(), ...), float -> int
  if (k < 0) return best;                        // edlib: k < 0 = unbounded
  return best <= min(m, k) ? best : -1;
}

// score of _editDistanceHW, src/coverage.h:112
__device__ __forceinline__ double cls_score(int dist, int m, float fq) {
  if (dist == -1) return 0.0;
  return ((1.0 - (double)fq) * (double)m) / (double)(dist + 1);
}

// One text column for NP probes in lock-step (P/M/score/best: the per-probe Myers states).  WT = the word type:
// uint64_t in general, uint32_t when every probe of the wavefront fits 32 rows (half the VALU work).
template <typename WT, int NW, int NP, bool SLOW>
__device__ __forceinline__ void cls_column(const ClsLds<NW>& L, int c, bool valid, int lane, int p0,
                                           const uint8_t* const (&probe)[2], const int (&m)[2], WT (&P)[NP][NW],
                                           WT (&M)[NP][NW], int (&score)[NP], int (&best)[NP]) {
  constexpr int WB = (int)sizeof(WT) * 8;
  static_assert(WB == 64 || NW == 1, "32-bit words: one-word probes only");
  const int ci = (int)L.lut[c] + lane;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int mm = m[p0 + p];
    const int wl = (mm - 1) >> 6, bl = (mm - 1) & (WB - 1);
    int hin = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int slot = ((p0 + p) * NW + w) * CLS_NCODE * WAVE + ci;
      WT Eq = (WB == 64) ? (WT)L.peq[slot] : (WT) reinterpret_cast<const uint32_t*>(L.peq)[2 * slot];
      if (SLOW) {
        if (cls_code(c) == 5) Eq = (WT)cls_eq_slow(probe[p0 + p], mm, w, c);
      }
      const WT Pv = P[p][w], Mv = M[p][w];
      const WT hinNeg = (NW > 1 && hin < 0) ? 1 : 0;
      const WT Xv = Eq | Mv;
      Eq |= hinNeg;
      const WT Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
      WT Ph = Mv | ~(Xh | Pv);
      WT Mh = Pv & Xh;
      if (NW == 1 || w == wl) score[p] += (int)((Ph >> bl) & 1) - (int)((Mh >> bl) & 1);
      const int hout = (int)(Ph >> (WB - 1)) - (int)(Mh >> (WB - 1));
      Ph <<= 1;
      Mh <<= 1;
      if (NW > 1) {
        Mh |= hinNeg;
        Ph |= (hin > 0) ? 1 : 0;
      }
      P[p][w] = Mh | ~(Xv | Ph);
      M[p][w] = Ph & Xv;
      hin = hout;
    }
    const int nb = min(best[p], score[p]);
    best[p] = valid ? nb : best[p];
  }
}

// Scans the read.  Chunks of 16 columns that every lane of the wavefront owns run unrolled and unmasked
// (one-word probes without foreign bytes); ragged ends, wide probes and probes with bytes outside ACGTN
// take the rolled loop.
template <typename WT, int NW, int NP, bool SLOW>
__device__ __forceinline__ void cls_scan(const ClsLds<NW>& L, const uint8_t* seq, int n, int nmin, int nmax, int lane, int p0,
                                         const uint8_t* const (&probe)[2], const int (&m)[2], int (&best)[NP]) {
  WT P[NP][NW], M[NP][NW];
  int score[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      P[p][w] = ~(WT)0;
      M[p][w] = 0;
    }
    score[p] = m[p0 + p];
    best[p] = m[p0 + p];
  }
  uint4 cur = cls_load16(seq);
  int base = 0;
  if (!SLOW && NW == 1) {
    for (; base + 16 <= nmin; base += 16) {
      const uint4 nxt = cls_load16(seq + min(base + 16, n));
      const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (int f = 0; f < 16; ++f)
        cls_column<WT, NW, NP, false>(L, (int)((wd[f >> 2] >> ((f & 3) * 8)) & 0xff), true, lane, p0, probe, m, P, M, score, best);
      cur = nxt;
    }
  }
  for (; base < nmax; base += 16) {
    const uint4 nxt = cls_load16(seq + min(base + 16, n));
    uint64_t lo = (uint64_t)cur.x | ((uint64_t)cur.y << 32), hi = (uint64_t)cur.z | ((uint64_t)cur.w << 32);
    const int fend = min(16, nmax - base);
#pragma unroll 1
    for (int f = 0; f < fend; ++f) {
      cls_column<WT, NW, NP, SLOW>(L, (int)(lo & 0xff), base + f < n, lane, p0, probe, m, P, M, score, best);
      lo = (lo >> 8) | (hi << 56);
      hi >>= 8;
    }
    cur = nxt;
  }
}

template <int NW>
__global__ __launch_bounds__(WAVE) void classify_kernel(ClsArgs A) {
  __shared__ ClsLds<NW> L;
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += WAVE) L.lut[i] = (uint16_t)(cls_code(i) * WAVE);
  __syncthreads();   // one wavefront per block; everything else in LDS is private to its lane
  const uint64_t n_items = (NW == 1) ? A.n_jobs : (uint64_t)(*A.wide_count);
  const uint64_t n_groups = (n_items + WAVE - 1) / WAVE;
  for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint64_t item = g * WAVE + lane;
    bool active = item < n_items;
    const uint64_t idx = !active ? 0 : (NW == 1 ? item : (uint64_t)A.wide_list[item]);
    dellyhip_align_job J{};
    if (active) J = A.jobs[idx];
    int m[2] = {(int)J.cons_len, (int)J.ref_len};
    int n = (int)J.seq_len;
    const uint8_t* const probe[2] = {A.blob + J.cons_off, A.blob + J.ref_off};
    const uint8_t* seq = A.blob + J.seq_off;
    bool limit = false;
    if (active && max(m[0], m[1]) > 64 * NW) {
      if (NW == 1) {   // hand over to the wide launch
        const int slot = atomicAdd(A.wide_count, 1);
        A.wide_list[slot] = (int32_t)idx;
      } else if (max(m[0], m[1]) <= MYERS_ROWS) {   // ... to the one-job-per-wavefront launch
        const int slot = atomicAdd(A.big_count, 1);
        A.big_list[slot] = (int32_t)idx;
      } else {
        limit = true;
      }
      active = false;
    }
    if (!active) {
      m[0] = m[1] = 0;
      n = 0;
    }
    // equality masks of both probes, accumulated in this lane's LDS slots (slot 5 = bytes outside ACGTN: cleared
    // afterwards, it doubles as the all-zero mask of such read bytes)
    bool other = false;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        uint64_t* slot = &L.peq[(p * NW + w) * CLS_NCODE * WAVE + lane];
#pragma unroll
        for (int k = 0; k < CLS_NCODE; ++k) slot[k * WAVE] = 0;
        const int rend = min(m[p], w * 64 + 64);
        for (int r0 = w * 64; r0 < rend; r0 += 8) {
          uint64_t v;
          __builtin_memcpy(&v, probe[p] + r0, 8);
          const int bend = min(8, rend - r0);
          for (int b = 0; b < bend; ++b) {
            const int code = (int)L.lut[(int)(v & 0xff)];
            v >>= 8;
            other |= code == 5 * WAVE;
            slot[code] |= 1ull << ((r0 + b) & 63);
          }
        }
        slot[5 * WAVE] = 0;
      }
    }
    const int nmax = cls_wave_max(n);
    const int nmin = min(nmax, -cls_wave_max(active ? -n : -0x7fffffff));   // no active lane: 0
    const bool any_other = __ballot(other) != 0;
    int best[2] = {0, 0};
    if (NW == 1) {
      const int mmax = cls_wave_max(max(m[0], m[1]));
      if (any_other) cls_scan<uint64_t, NW, 2, true>(L, seq, n, nmin, nmax, lane, 0, probe, m, best);
      else if (mmax <= 32) cls_scan<uint32_t, 1, 2, false>(reinterpret_cast<const ClsLds<1>&>(L), seq, n, nmin, nmax, lane, 0, probe, m, best);
      else cls_scan<uint64_t, NW, 2, false>(L, seq, n, nmin, nmax, lane, 0, probe, m, best);
    } else {
      int b1[1];
      cls_scan<uint64_t, NW, 1, true>(L, seq, n, nmin, nmax, lane, 0, probe, m, b1);
      best[0] = b1[0];
      cls_scan<uint64_t, NW, 1, true>(L, seq, n, nmin, nmax, lane, 1, probe, m, b1);
      best[1] = b1[0];
    }
    if (active || limit) {
      dellyhip_align_result R{};
      R.type = 'N';
      if (limit) {
        R.status = DELLYHIP_E_LIMIT;
        R.dist_alt = R.dist_ref = -1;
      } else {
        const float fq = A.flank_quality;
        R.dist_alt = cls_distance(best[0], m[0], n, fq);
        R.dist_ref = cls_distance(best[1], m[1], n, fq);
        const double scoreAlt = cls_score(R.dist_alt, m[0], fq);
        const double scoreRef = cls_score(R.dist_ref, m[1], fq);
        if (scoreRef > 0.7 || scoreAlt > 0.7) {   // src/coverage.h:424-433
          R.sv_id = J.sv_id;
          R.file_index = J.file_index;
          if (scoreRef > scoreAlt) {
            R.type = 'R';
            R.qual = (uint8_t)min(255, min((int)(scoreRef * 35), (int)J.qual));
          } else {
            R.type = 'A';
            R.qual = (uint8_t)min(255, min((int)(scoreAlt * 35), (int)J.qual));
          }
        }
      }
      A.res[idx] = R;
    }
  }
}

// probes of 257 .. 6144 bytes (a consensus-sized homology; practically never): one job per wavefront, the probe's rows cut
// into 32-row words over the lanes (myers_hw_distance), exact byte comparison
__global__ __launch_bounds__(WAVE) void classify_big_kernel(ClsArgs A) {
  const int lane = threadIdx.x;
  const int n_items = __builtin_amdgcn_readfirstlane(*A.big_count);
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int idx = A.big_list[it];
    const dellyhip_align_job J = A.jobs[idx];
    const int m[2] = {(int)J.cons_len, (int)J.ref_len};
    const int n = (int)J.seq_len;
    const uint8_t* const probe[2] = {A.blob + J.cons_off, A.blob + J.ref_off};
    const uint8_t* seq = A.blob + J.seq_off;
    int best[2];
    for (int p = 0; p < 2; ++p) {
      if (m[p] == 0 || n == 0) best[p] = m[p];
      else if (m[p] <= WAVE * 32) best[p] = myers_hw_distance<1>(probe[p], m[p], seq, n, lane);
      else if (m[p] <= WAVE * 64) best[p] = myers_hw_distance<2>(probe[p], m[p], seq, n, lane);
      else best[p] = myers_hw_distance<3>(probe[p], m[p], seq, n, lane);
    }
    if (lane == 0) {
      dellyhip_align_result R{};
      R.type = 'N';
      const float fq = A.flank_quality;
      R.dist_alt = cls_distance(best[0], m[0], n, fq);
      R.dist_ref = cls_distance(best[1], m[1], n, fq);
      const double scoreAlt = cls_score(R.dist_alt, m[0], fq);
      const double scoreRef = cls_score(R.dist_ref, m[1], fq);
      if (scoreRef > 0.7 || scoreAlt > 0.7) {   // src/coverage.h:424-433
        R.sv_id = J.sv_id;
        R.file_index = J.file_index;
        if (scoreRef > scoreAlt) {
          R.type = 'R';
          R.qual = (uint8_t)min(255, min((int)(scoreRef * 35), (int)J.qual));
        } else {
          R.type = 'A';
          R.qual = (uint8_t)min(255, min((int)(scoreAlt * 35), (int)J.qual));
        }
      }
      A.res[idx] = R;
    }
  }
}

}  // namespace dh
