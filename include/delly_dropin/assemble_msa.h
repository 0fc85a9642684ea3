/* assemble_msa.h -- drop-in for the long-read consensus entry points of the reference's src/assemble.h:
 *   template<TConfig, TSplitReadSet> int msaEdlib(TConfig const& c, TSplitReadSet& sps, std::string& cs)             (:383-473)
 *   template<TConfig, TSplitReadSet> int msaWfa(TConfig const& c, TSplitReadSet& sps, std::string& cs,
 *                                               std::string const& prefix, std::string const& suffix)                (:547-726)
 *   template<TConfig, TSplitReadSet> int msaWfa(TConfig const& c, TSplitReadSet& sps, std::string& cs)               (:728-732)
 * (assemble() itself, the BAM loop of :734-964, stays the reference's; its loop body :836-861 is
 * dellyhip_refine_batch_lr when batched). */
#ifndef DELLYHIP_DROPIN_ASSEMBLE_MSA_H
#define DELLYHIP_DROPIN_ASSEMBLE_MSA_H

#include "dellyhip_dropin.h"

namespace torali {

namespace dellyhip_detail {
template <typename TConfig>
inline dellyhip_params lr_params(TConfig const& c) {
  dellyhip_params d;
  dellyhip_default_params_lr(&d);   // fields a long-read config may lack (aliscore) keep the `delly lr` defaults
  dellyhip_params p = dellyhip_dropin::make_params(c);
  if (!dellyhip_dropin::has_aliscore<TConfig>::value) { p.match = d.match; p.mismatch = d.mismatch; p.gap_open = d.gap_open; p.gap_extend = d.gap_extend; }
  if (!dellyhip_dropin::has_minCliqueSize<TConfig>::value) p.min_clique_size = d.min_clique_size;
  if (!dellyhip_dropin::has_flankQuality<TConfig>::value) p.flank_quality = d.flank_quality;
  if (!dellyhip_dropin::has_minimumFlankSize<TConfig>::value) p.minimum_flank_size = d.minimum_flank_size;
  if (!dellyhip_dropin::has_indelsize<TConfig>::value) p.indelsize = d.indelsize;
  if (!dellyhip_dropin::has_minConsWindow<TConfig>::value) p.min_cons_window = d.min_cons_window;
  return p;
}
}  // namespace dellyhip_detail

template <typename TConfig, typename TSplitReadSet>
inline int msaEdlib(TConfig const& c, TSplitReadSet& sps, std::string& cs) {
  namespace dd = dellyhip_dropin;
  cs.clear();
  if (sps.size() == 0) return 0;
  std::string blob;
  std::vector<uint64_t> off(1, 0);
  dd::pack_reads(sps, blob, off);
  std::size_t longest = 0;
  for (std::size_t i = 0; i + 1 < off.size(); ++i) longest = std::max<std::size_t>(longest, off[i + 1] - off[i]);
  dd::Session& S = dd::session(dellyhip_detail::lr_params(c));
  std::vector<char> out(2 * longest + 4096);
  int32_t len = 0, rows = 0;
  dd::check(dellyhip_msa_edlib(S.ctx, (int32_t)sps.size(), blob.data(), off.data(), out.data(), (int32_t)out.size(), &len, &rows));
  cs.assign(out.data(), (std::size_t)len);
  return rows;
}

template <typename TConfig, typename TSplitReadSet>
inline int msaWfa(TConfig const& c, TSplitReadSet& sps, std::string& cs, std::string const& prefix, std::string const& suffix) {
  namespace dd = dellyhip_dropin;
  cs.clear();
  if (sps.size() == 0) return 0;
  std::string blob;
  std::vector<uint64_t> off(1, 0);
  dd::pack_reads(sps, blob, off);
  std::size_t longest = 0;
  for (std::size_t i = 0; i + 1 < off.size(); ++i) longest = std::max<std::size_t>(longest, off[i + 1] - off[i]);
  dd::Session& S = dd::session(dellyhip_detail::lr_params(c));
  std::vector<char> out(2 * longest + 4096);
  int32_t len = 0, rows = 0;
  dd::check(dellyhip_msa_wfa(S.ctx, (int32_t)sps.size(), blob.data(), off.data(), prefix.data(), (int32_t)prefix.size(), suffix.data(),
                             (int32_t)suffix.size(), out.data(), (int32_t)out.size(), &len, &rows));
  cs.assign(out.data(), (std::size_t)len);
  return rows;
}

template <typename TConfig, typename TSplitReadSet>
inline int msaWfa(TConfig const& c, TSplitReadSet& sps, std::string& cs) {
  return msaWfa(c, sps, cs, "", "");
}

}  // namespace torali

#endif
