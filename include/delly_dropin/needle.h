/* needle.h -- drop-in for the entry point of the reference's src/needle.h the path uses:
 *   template<TAlign, TAlignConfig, TScoreObject>
 *   bool longNeedle(std::string const& s1, std::string const& s2, TAlign& align, TAlignConfig const& ac, TScoreObject const& sc)
 * (src/needle.h:45-222) exactly as _consRefAlignment calls it (src/split.h:543-555): AlignConfig<true, false>,
 * DnaScore(1, -1, -1, -1).  Any other configuration has no caller in the reference and is rejected.
 * Needs the reference's align.h (AlignConfig, DnaScore) and boost::multi_array for TAlign. */
#ifndef DELLYHIP_DROPIN_NEEDLE_H
#define DELLYHIP_DROPIN_NEEDLE_H

#include <boost/multi_array.hpp>

#include "align.h"
#include "dellyhip_dropin.h"

namespace torali {

template <typename TAlign, typename TAlignConfig, typename TScoreObject>
inline bool longNeedle(std::string const& s1, std::string const& s2, TAlign& align, TAlignConfig const&, TScoreObject const& sc) {
  namespace dd = dellyhip_dropin;
  static_assert(std::is_same<TAlignConfig, AlignConfig<true, false> >::value,
                "dellyhip longNeedle: only AlignConfig<true,false> (src/split.h:543) is built");
  if (!(sc.match == 1 && sc.mismatch == -1 && sc.go == -1 && sc.ge == -1))
    throw dd::Error(DELLYHIP_E_ARG, "longNeedle: only DnaScore(1,-1,-1,-1) (src/split.h:544) is built");
  dellyhip_params p;
  dellyhip_default_params_sr(&p);
  dd::Session& S = dd::session(p);
  const int32_t cap = (int32_t)(s1.size() + s2.size() + 8);
  std::vector<char> rows(2 * (std::size_t)cap);
  int32_t len = 0, found = 0;
  dd::check(dellyhip_long_needle(S.ctx, s1.data(), (int32_t)s1.size(), s2.data(), (int32_t)s2.size(), rows.data(), cap, &len, &found));
  if (!found) return false;   // (the reference leaves `align` as it is, src/needle.h:83-85,152)
  align.resize(boost::extents[2][len]);
  for (int32_t j = 0; j < len; ++j) {
    align[0][j] = rows[j];
    align[1][j] = rows[(std::size_t)cap + j];
  }
  return true;
}

}  // namespace torali

#endif
