/* gotoh.h -- drop-in for the reference's src/gotoh.h:
 *   template<TAlign1, TAlign2, TAlign, TAlignConfig, TScoreObject>
 *   int gotoh(TAlign1 const& a1, TAlign2 const& a2, TAlign& align, TAlignConfig const& ac, TScoreObject const& sc)  (src/gotoh.h:71-174)
 * as palign calls it (src/msa.h:106-107): AlignConfig<true, true>; any DnaScore.  The three-argument overload of the
 * reference uses AlignConfig<false,false> (src/gotoh.h:185-189) and has no caller: not built.
 * Needs the reference's align.h and boost::multi_array. */
#ifndef DELLYHIP_DROPIN_GOTOH_H
#define DELLYHIP_DROPIN_GOTOH_H

#include <boost/multi_array.hpp>

#include "align.h"
#include "dellyhip_dropin.h"

namespace torali {

template <typename TAlign1, typename TAlign2, typename TAlign, typename TAlignConfig, typename TScoreObject>
inline int gotoh(TAlign1 const& a1, TAlign2 const& a2, TAlign& align, TAlignConfig const&, TScoreObject const& sc) {
  namespace dd = dellyhip_dropin;
  static_assert(std::is_same<TAlignConfig, AlignConfig<true, true> >::value,
                "dellyhip gotoh: only AlignConfig<true,true> (src/msa.h:106) is built");
  const int32_t r1 = (int32_t)a1.shape()[0], m = (int32_t)a1.shape()[1];
  const int32_t r2 = (int32_t)a2.shape()[0], n = (int32_t)a2.shape()[1];
  std::string f1((std::size_t)r1 * m, '-'), f2((std::size_t)r2 * n, '-');
  for (int32_t i = 0; i < r1; ++i)
    for (int32_t j = 0; j < m; ++j) f1[(std::size_t)i * m + j] = a1[i][j];
  for (int32_t i = 0; i < r2; ++i)
    for (int32_t j = 0; j < n; ++j) f2[(std::size_t)i * n + j] = a2[i][j];
  dellyhip_params p;
  dellyhip_default_params_sr(&p);
  p.match = (int32_t)sc.match;
  p.mismatch = (int32_t)sc.mismatch;
  p.gap_open = (int32_t)sc.go;
  p.gap_extend = (int32_t)sc.ge;
  dd::Session& S = dd::session(p);
  const int32_t cap = m + n + 8;
  std::vector<char> out((std::size_t)(r1 + r2) * cap);
  int32_t len = 0, score = 0;
  dd::check(dellyhip_gotoh(S.ctx, f1.data(), r1, m, f2.data(), r2, n, out.data(), cap, &len, &score));
  align.resize(boost::extents[r1 + r2][len]);
  for (int32_t i = 0; i < r1 + r2; ++i)
    for (int32_t j = 0; j < len; ++j) align[i][j] = out[(std::size_t)i * cap + j];
  return score;
}

template <typename TAlign1, typename TAlign2, typename TAlign, typename TAlignConfig>
inline int gotoh(TAlign1 const& a1, TAlign2 const& a2, TAlign& align, TAlignConfig const& ac) {
  DnaScore<int> dnasc;   // src/gotoh.h:176-183
  return gotoh(a1, a2, align, ac, dnasc);
}

}  // namespace torali

#endif
