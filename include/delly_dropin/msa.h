/* msa.h -- drop-in for the entry point of the reference's src/msa.h:
 *   template<TConfig, TSplitReadSet> int msa(TConfig const& c, TSplitReadSet const& sps, std::string& cs)   (src/msa.h:185-239)
 * lcs / distanceMatrix / upgma / palign / gotoh / consensus run inside one MI355X kernel (delly_amd/csrc/msa_kernel.hpp). */
#ifndef DELLYHIP_DROPIN_MSA_H
#define DELLYHIP_DROPIN_MSA_H

#include "dellyhip_dropin.h"

namespace torali {

template <typename TConfig, typename TSplitReadSet>
inline int msa(TConfig const& c, TSplitReadSet const& sps, std::string& cs) {
  namespace dd = dellyhip_dropin;
  // (the reference never clears cs: consensus() appends with push_back, src/msa.h:170-172 -- so does this)
  const std::size_t n = sps.size();
  if (n == 0) return 0;
  if (n == 1) return 1;   // one row, coverage below max(2, ...) everywhere: empty consensus (src/msa.h:111-173)
  std::string blob;
  std::vector<uint64_t> off(1, 0);
  dd::pack_reads(sps, blob, off);
  std::size_t longest = 0;
  for (std::size_t i = 0; i + 1 < off.size(); ++i) longest = std::max<std::size_t>(longest, off[i + 1] - off[i]);
  dd::Session& S = dd::session(dd::make_params(c));
  std::vector<char> out(2 * longest + 2048);
  int32_t len = 0, rows = 0;
  dd::check(dellyhip_msa(S.ctx, (int32_t)n, blob.data(), off.data(), out.data(), (int32_t)out.size(), &len, &rows));
  cs.append(out.data(), (std::size_t)len);
  return rows;
}

}  // namespace torali

#endif
