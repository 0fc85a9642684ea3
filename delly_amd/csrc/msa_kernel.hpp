// msa_kernel.hpp -- gfx950 device code for msa() (src/msa.h:185-239):
// pairwise LCS distance matrix, UPGMA guide tree, progressive profile Gotoh,
// consensus vote.   (stage under construction: entry points report
// DELLYHIP_E_LIMIT until the kernels land)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/dellyhip.h"

namespace dh {

struct MsaArgs {
  const dellyhip_junction* junc;
  const uint8_t* seq_blob;
  const uint64_t* seq_off;
  dellyhip_params p;
  dellyhip_result* res;
  uint8_t* out_blob;
  uint64_t out_stride;
  int32_t* cons_len;
  uint8_t* ws;
  uint64_t ws_stride;
  int32_t n_work;
  int32_t* work_counter;
};

inline int msa_prepare(const std::vector<dellyhip_junction>&, const uint64_t*, uint64_t& ws_stride) {
  ws_stride = 0;
  return DELLYHIP_E_LIMIT;
}
inline int msa_launch(const MsaArgs&, int, hipStream_t) { return DELLYHIP_E_LIMIT; }
inline int msa_single_lcs(hipStream_t, const char*, int, const char*, int, int32_t*) { return DELLYHIP_E_LIMIT; }
inline int msa_single_gotoh(hipStream_t, const dellyhip_params&, const char*, int, int, const char*, int, int, char*,
                            int, int32_t*, int32_t*) { return DELLYHIP_E_LIMIT; }
inline int msa_single(hipStream_t, const dellyhip_params&, int, int, const char*, const uint64_t*, char*, int,
                      int32_t*, int32_t*) { return DELLYHIP_E_LIMIT; }

}  // namespace dh
