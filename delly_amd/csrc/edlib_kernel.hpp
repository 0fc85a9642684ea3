// edlib_kernel.hpp -- gfx950 device code behind dellyhip_edlib_align_full: ONE edlibAlign call (src/edlib.h:242-246,
// src/edlib.cpp:139-300) of any shape the strip machinery takes (target <= 32 766 letters, query <= 32 000), with everything
// the result struct holds: editDistance, ALL optimal end locations (ascending, as myersCalcEditDistanceSemiGlobal collects
// them, src/edlib.cpp:653-691), the start location of each (HW: last optimal end of the reversed SHW sweep over the target
// prefix, :236-249), and the alignment of the first pair (obtainAlignment incl. its Hirschberg regime, :1163-1389) -- with
// or without the 20 extended-IUPAC equality pairs of msaEdlib / msaWfa (src/assemble.h:425).  One wavefront; the primitives
// are those of lrmsa_kernel.hpp (lm_locate, lm_hw, lm_shw, lm_nw_path), which the long-read kernels are tested through.
#pragma once
#include "lrmsa_kernel.hpp"

namespace dh {

struct EdFullArgs {
  const uint8_t* q;        // query   (edlib's first sequence)
  const uint8_t* t;        // target
  int32_t qn, tn;
  int32_t mode;            // 0 NW, 1 SHW, 2 HW
  int32_t task;            // 0 DISTANCE, 1 LOC, 2 PATH
  int32_t eq;              // 1: the extended-IUPAC equalities
  uint8_t* ws;             // workspace: boundary rows | tmp | last column | directions
  uint64_t off_tmp, off_lastcol, off_dirs;
  int32_t bnd_stride;
  uint64_t strip_words;
  int32_t* out;            // {editDistance, numLocations, alignmentLength, status (0 / DELLYHIP_E_LIMIT)}
  int32_t* end_locs;       // numLocations entries (<= loc_cap)
  int32_t* start_locs;     // ... (LOC / PATH)
  int32_t loc_cap;
  uint8_t* ops;            // alignment, forward order
  int32_t ops_cap;
};

__global__ __launch_bounds__(WAVE) void edlib_full_kernel(EdFullArgs a) {
  const int lane = threadIdx.x;
  const int qn = a.qn, tn = a.tn;
  int32_t* bnd = reinterpret_cast<int32_t*>(a.ws);
  uint8_t* tmp = a.ws + a.off_tmp;
  int32_t* lastcol = reinterpret_cast<int32_t*>(a.ws + a.off_lastcol);
  uint32_t* dirs = reinterpret_cast<uint32_t*>(a.ws + a.off_dirs);
  int emode = 0;
  if (a.eq) emode = LM_EQ | ((lm_in_classes(a.t, 1, tn, lane) && lm_in_classes(a.q, 1, qn, lane)) ? LM_EQFAST : 0);
  const bool loc = a.task >= 1, path = a.task >= 2;
  int ed = 0, nloc = 0, nops = 0, status = 0;
  if (a.mode == 0) {   // NW: the single admissible end is the last target letter
    nloc = 1;
    if (lane == 0) { a.end_locs[0] = tn - 1; if (loc) a.start_locs[0] = 0; }
    if (path) {
      nops = lm_nw_path(a.t, tn, a.q, qn, emode, bnd, a.bnd_stride, dirs, a.strip_words, tmp, a.ops, a.ops_cap, lane);
      if (nops < 0) { status = DELLYHIP_E_LIMIT; nops = 0; }
      int e = 0;
      for (int k = lane; k < nops; k += WAVE) e += (a.ops[k] != (uint8_t)ED_MATCH) ? 1 : 0;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) e += __shfl_xor(e, o);
      ed = rfl(e);
    } else {
      int32_t* row = bnd + 2 * a.bnd_stride;
      lm_last_row(a.t, 1, tn, a.q, 1, qn, emode, bnd, bnd + a.bnd_stride, row, lane);
      ed = rfl((int)__hip_atomic_load(row + qn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  } else {
    const int hw = (a.mode == 2) ? LM_HW : 0;
    int first, last;
    lm_locate(a.t, 1, tn, a.q, 1, qn, emode | hw, bnd, bnd + a.bnd_stride, lane, ed, first, last, lastcol);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // every optimal end, ascending (row r of the last column = end location r - 1)
    const int r0 = lm_first_row(qn);
    for (int base = r0; base <= tn; base += WAVE) {
      const int r = base + lane;
      const bool hit = (r <= tn) && ((int)__hip_atomic_load(lastcol + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ed);
      const unsigned long long bm = __ballot(hit);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pos = nloc + __popcll(bm & below);
      if (hit && pos < a.loc_cap) a.end_locs[pos] = r - 1;
      nloc += __popcll(bm);
    }
    if (nloc > a.loc_cap) { status = DELLYHIP_E_LIMIT; nloc = a.loc_cap; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int end0 = first - 1;
    int start0 = 0;
    if (loc) {
      for (int i = 0; i < nloc; ++i) {
        int st = 0;
        const int e = rfl((int)__hip_atomic_load(a.end_locs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (hw && e >= 0) {   // src/edlib.cpp:236-249: the LAST optimal end of the reversed SHW sweep over target[0 .. e]
          int ed2, f2, l2;
          lm_locate(a.t + e, -1, e + 1, a.q + (qn - 1), -1, qn, emode, bnd, bnd + a.bnd_stride, lane, ed2, f2, l2);
          st = e - (l2 - 1);
        }
        if (lane == 0) a.start_locs[i] = st;
        if (i == 0) start0 = st;
      }
    }
    if (path && nloc > 0) {
      const int tl = end0 - start0 + 1;
      if (end0 < 0 || tl <= 0) nops = lm_fill_inserts(a.ops, qn, lane);     // src/edlib.cpp:222-235,1169-1176
      else nops = lm_nw_path(a.t + start0, tl, a.q, qn, emode, bnd, a.bnd_stride, dirs, a.strip_words, tmp, a.ops, a.ops_cap, lane);
      if (nops < 0) { status = DELLYHIP_E_LIMIT; nops = 0; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) {
    a.out[0] = ed;
    a.out[1] = nloc;
    a.out[2] = nops;
    a.out[3] = status;
  }
}

}  // namespace dh
