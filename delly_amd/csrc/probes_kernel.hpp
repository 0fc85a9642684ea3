// probes_kernel.hpp -- gfx950 device code: the per-SV part of _generateProbes (src/coverage.h:164-263)
// behind a refinement batch.  _generateProbes re-runs _consRefAlignment + _findSplit per precise SV
// (:214-217); here the alignment descriptor (cStart, cEnd, rStart, rEnd, homLeft, homRight) is the one
// the split kernels left in the result record, so only the four substrings (:255-256) and the BpRegion
// fields (:236-253,257) remain: one junction per wavefront, svRefStr bytes re-derived from the
// chromosome through the same segment list as the alignment (window_segments).
#pragma once
#include "split_main.hpp"

namespace dh {

constexpr int PROBE_CAP = 1024;   // bytes per probe slot: probe = 2*minimumFlankSize + homLeft + homRight <= 2*13 + 2*319 on short-read shapes

struct ProbeArgs {
  SplitArgs a;
  dellyhip_probes* out;
  uint8_t* blob;        // 4 * PROBE_CAP bytes per junction: consProbe[0], refProbe[0], consProbe[1], refProbe[1]
  int32_t n;
};

// byte i of svRefStr = the concatenated (reverse-complemented, upper-cased) segments (split.h:70-163)
__device__ __forceinline__ uint8_t window_char(const Seg (&seg)[3], int nseg, int i) {
  for (int q = 0; q < nseg; ++q) {
    const Seg& sg = seg[q];
    if (i < sg.len) {
      const uint8_t fwd = upc(sg.base[sg.beg + i]);
      if (!sg.rc) return fwd;
      const uint8_t r = comp_acgtn(upc(sg.base[sg.beg + (sg.len - 1 - i)]));
      return r ? r : fwd;
    }
    i -= sg.len;
  }
  return 0;
}

// _cutRefStart / _cutRefEnd  src/coverage.h:117-162: which descriptor end the breakpoint sits on
__device__ __forceinline__ int cut_ref_anchor(int rStart, int rEnd, int bpPoint, int svt) {
  const int ct = is_tra(svt) ? svt - 5 : svt;       // _getSpanOrientation, src/tags.h:33-40
  const bool flipped = (ct == 3);                   // (non-translocations: svt == 3)
  const bool use_end = flipped ? (bpPoint == 0) : (bpPoint != 0);
  return use_end ? rEnd : rStart;
}

__global__ __launch_bounds__(WAVE) void probes_kernel(ProbeArgs A) {
  const int lane = threadIdx.x;
  const SplitArgs& a = A.a;
  const dellyhip_params& P = a.p;
  for (int j = blockIdx.x; j < A.n; j += gridDim.x) {
    const dellyhip_junction J = a.junc[j];
    const dellyhip_result R = a.res[j];
    const int m = a.cons_len[j];
    dellyhip_probes O{};
    O.svid = J.svid;
    O.status = R.status;
    // src/coverage.h:236-253 (independent of the alignment)
    const int mfs = P.minimum_flank_size;
    O.region_start[0] = max(0, J.sv_start - mfs);
    O.region_end[0] = (int32_t)min((uint32_t)(J.sv_start + mfs), (uint32_t)a.chr_len[J.chr]);
    O.bppos[0] = J.sv_start;
    O.region_start[1] = max(0, J.sv_end - mfs);
    O.region_end[1] = (int32_t)min((uint32_t)(J.sv_end + mfs), (uint32_t)a.chr_len[J.chr2]);
    O.bppos[1] = J.sv_end;
    bool ok = R.status == 0 && R.r_end > R.r_start;   // the descriptor is only written when _findSplit succeeded
    Seg seg[3];
    int nseg = 0, n = 0, sBeg, sEnd, eBeg, eEnd;
    if (ok) {
      const bool built = (J.svt == 4) ? window_segments<true>(a, J, m, seg, nseg, sBeg, sEnd, eBeg, eEnd)
                                      : window_segments<false>(a, J, m, seg, nseg, sBeg, sEnd, eBeg, eEnd);
      ok = built;
      for (int q = 0; q < nseg; ++q) n += seg[q].len;
      if (m > MMAX || n > NMAX) {   // long-read shapes: the long-read genotyper (src/genotype.h) uses no probes
        ok = false;
        O.status = DELLYHIP_E_LIMIT;
      }
    }
    if (ok) {
      const uint8_t* cons = a.cons_base + a.cons_off[j];
      uint8_t* slot = A.blob + (size_t)j * 4 * PROBE_CAP;
      O.hom_left = R.hom_left;
      O.hom_right = R.hom_right;
      for (int bp = 0; bp < 2; ++bp) {
        const int cAnchor = bp ? R.c_end : R.c_start;
        const int cs = cAnchor - R.hom_left - mfs, ce = cAnchor + R.hom_right + mfs;
        const int rAnchor = cut_ref_anchor(R.r_start, R.r_end, bp, J.svt);
        const int rs = rAnchor - (R.hom_left + mfs), re = rAnchor + (R.hom_right + mfs);
        // std::string::substr(pos, len): pos must lie in [0, size] (guaranteed by the flank tests of _findSplit,
        // src/split.h:370-371), len is clipped at the end of the string
        if (cs < 0 || cs > m || rs < 0 || rs > n) {
          ok = false;
          O.status = DELLYHIP_E_LIMIT;
          break;
        }
        const int cl = min(ce - cs, m - cs), rl = min(re - rs, n - rs);
        if (cl > PROBE_CAP || rl > PROBE_CAP) {
          ok = false;
          O.status = DELLYHIP_E_LIMIT;
          break;
        }
        uint8_t* cdst = slot + (2 * bp) * PROBE_CAP;
        uint8_t* rdst = slot + (2 * bp + 1) * PROBE_CAP;
        for (int i = lane; i < cl; i += WAVE) cdst[i] = cons[cs + i];
        for (int i = lane; i < rl; i += WAVE) rdst[i] = window_char(seg, nseg, rs + i);
        O.cons_len[bp] = cl;
        O.ref_len[bp] = rl;
        O.cons_off[bp] = (uint64_t)j * 4 * PROBE_CAP + (uint64_t)(2 * bp) * PROBE_CAP;
        O.ref_off[bp] = (uint64_t)j * 4 * PROBE_CAP + (uint64_t)(2 * bp + 1) * PROBE_CAP;
      }
    }
    if (!ok) {
      O.hom_left = O.hom_right = 0;
      for (int bp = 0; bp < 2; ++bp) {
        O.cons_len[bp] = O.ref_len[bp] = 0;
        O.cons_off[bp] = O.ref_off[bp] = 0;
      }
    }
    O.ok = ok ? 1 : 0;
    if (lane == 0) A.out[j] = O;
  }
}

}  // namespace dh
