/* split.h -- drop-in for the entry points of the reference's src/split.h that the path calls from outside:
 *   template<TConfig> bool alignConsensus(TConfig const& c, bam_hdr_t const* hdr, char const* seq, char const* sndSeq,
 *                                         StructuralVariantRecord& sv, bool const realign)                  (src/split.h:644-666)
 *   template<TConfig> bool alignConsensus(TConfig const& c, bam_hdr_t* hdr, char const* seq, char const* sndSeq,
 *                                         StructuralVariantRecord& sv)                                      (:668-672)
 *   template<TAlign>  bool splitAlign(std::string const& cons, std::string const& svRefStr, TAlign& align)  (:480-538)
 *   template<TBPoint> void _adjustOrientation(std::string& sequence, TBPoint bpPoint, int32_t const svt)    (:55-68)
 * plus the batch seam the reference does not have:
 *   torali::refineBatch(c, hdr, seq, sndSeq, svs, seqStore, svidsToProcess)  == the loop of src/shortpe.h:175-201
 * _initBreakpoint / _getSVRef / _consRefAlignment / _findSplit / _percentIdentity / _findHomology / _coordTransform and
 * the exact alleles (src/split.h:70-375, :596-637) run inside the kernels; only their results come back.
 * Needs the reference's tags.h (StructuralVariantRecord, bam_hdr_t through htslib) and boost::multi_array. */
#ifndef DELLYHIP_DROPIN_SPLIT_H
#define DELLYHIP_DROPIN_SPLIT_H

#include <boost/multi_array.hpp>

#include "dellyhip_dropin.h"
#include "tags.h"

namespace torali {

/* src/split.h:55-68 (driven by src/shortpe.h:123-137 when the split reads are collected): reads of the far side of an
 * inversion / inverted translocation are reverse-complemented before the MSA.  _translocation(svt) is svt in 5..8 and
 * _getSpanOrientation(svt) is svt - 5 there (src/tags.h:22-40, src/util.h:238-248). */
template <typename TBPoint>
inline void _adjustOrientation(std::string& sequence, TBPoint bpPoint, int32_t const svt) {
  const bool far = bpPoint ? true : false;
  bool flip = false;
  if (svt >= 5 && svt <= 8) {
    const int ct = svt - 5;
    flip = (ct == 0 && far) || (ct == 1 && !far);
  } else if (svt == 0) flip = far;
  else if (svt == 1) flip = !far;
  if (flip) dellyhip_dropin::reverse_complement(sequence);
}

namespace dellyhip_detail {

/* what alignConsensus() writes into the record (src/split.h:606-637) */
inline void store_result(StructuralVariantRecord& sv, dellyhip_result const& r, const char* blob) {
  sv.precise = true;
  sv.svStart = r.sv_start;
  sv.svEnd = r.sv_end;
  sv.srAlignQuality = r.sr_align_quality;
  sv.insLen = r.ins_len;
  sv.consBp = r.cons_bp;
  sv.homLen = r.hom_len;
  sv.ciposlow = -r.ci_wiggle;
  sv.ciposhigh = r.ci_wiggle;
  sv.ciendlow = -r.ci_wiggle;
  sv.ciendhigh = r.ci_wiggle;
  if (r.allele_len > 0) sv.alleles.assign(blob + r.allele_off, (std::size_t)r.allele_len);
}

/* the same for a record of the compact payload (DELLYHIP_COMPACT_ALLELES: allele_len < 0, no allele bytes came back): "REF,ALT"
 * is cut from the caller's chromosome and the record's consensus exactly as src/split.h:606-624 does from the alignment --
 * REF = toupper(seq[svStartBeg + rStart - 1 .. + rEnd - 1)), ALT = consensus[cStart - 1 .. cEnd - 1).  `j` is the junction as it
 * was submitted (the record's coordinates before refinement place the window, src/tags.h:151-172). */
inline void store_result(StructuralVariantRecord& sv, dellyhip_result const& r, const char* blob, dellyhip_params const& p,
                         dellyhip_junction const& j, char const* seq, int64_t chr_len) {
  store_result(sv, r, blob);
  if (r.allele_len < 0) {
    sv.alleles.resize((std::size_t)(-(int64_t)r.allele_len));
    const int64_t got = dellyhip_recut_alleles(&p, &j, &r, blob + r.cons_off, seq, chr_len, &sv.alleles[0], sv.alleles.size());
    if (got != -(int64_t)r.allele_len) throw dellyhip_dropin::Error((int)std::min<int64_t>(got, DELLYHIP_E_ARG), "refineBatch: compact alleles could not be re-cut");
  }
}

inline dellyhip_junction junction_of(StructuralVariantRecord const& sv, int32_t tag, uint64_t seq_first, int32_t n_seq) {
  dellyhip_junction j;
  std::memset(&j, 0, sizeof j);
  j.svid = tag;
  j.svt = sv.svt;
  j.chr = sv.chr;
  j.chr2 = sv.chr2;
  j.sv_start = sv.svStart;
  j.sv_end = sv.svEnd;
  j.ins_len = sv.insLen;
  j.n_seq = n_seq;
  j.seq_first = seq_first;
  return j;
}

}  // namespace dellyhip_detail

template <typename TConfig>
inline bool alignConsensus(TConfig const& c, bam_hdr_t const* hdr, char const* seq, char const* sndSeq, StructuralVariantRecord& sv,
                           bool const realign) {
  namespace dd = dellyhip_dropin;
  dd::Session& S = dd::session(dd::make_params(c, realign));
  S.chromosome(sv.chr, seq, (int64_t)hdr->target_len[sv.chr]);
  if (sv.chr2 != sv.chr) {
    if (!sndSeq) throw dd::Error(DELLYHIP_E_ARG, "alignConsensus: chr2 != chr needs sndSeq (src/split.h:655)");
    S.chromosome(sv.chr2, sndSeq, (int64_t)hdr->target_len[sv.chr2]);
  }
  const dellyhip_junction j = dellyhip_detail::junction_of(sv, sv.id, 0, 1);
  const uint64_t off[2] = {0, (uint64_t)sv.consensus.size()};
  dellyhip_result r;
  std::vector<char> out(3 * sv.consensus.size() + (std::size_t)std::max(S.params.indelsize, 0) + 4096);
  uint64_t used = 0;
  dd::check(dellyhip_align_consensus_batch(S.ctx, 1, &j, sv.consensus.data(), off, 1, &r, out.data(), out.size(), &used, 0));
  if (r.status) throw dd::Error(r.status, "alignConsensus: junction beyond a kernel limit (INTEGRATION.md 5)");
  // the orientation test may have replaced the consensus by its reverse complement (src/split.h:564-572)
  if (realign && r.cons_len == (int32_t)sv.consensus.size() && r.cons_len > 0) sv.consensus.assign(out.data() + r.cons_off, (std::size_t)r.cons_len);
  if (!r.ok) return false;
  dellyhip_detail::store_result(sv, r, out.data());
  return true;
}

template <typename TConfig>
inline bool alignConsensus(TConfig const& c, bam_hdr_t* hdr, char const* seq, char const* sndSeq, StructuralVariantRecord& sv) {
  return alignConsensus(c, hdr, seq, sndSeq, sv, false);
}

template <typename TAlign>
inline bool splitAlign(std::string const& cons, std::string const& svRefStr, TAlign& align) {
  namespace dd = dellyhip_dropin;
  dellyhip_params p;
  dellyhip_default_params_sr(&p);
  dd::Session& S = dd::session(p);
  const int32_t cap = (int32_t)(cons.size() + svRefStr.size() + 8);
  std::vector<char> rows(2 * (std::size_t)cap);
  int32_t len = 0, found = 0;
  dd::check(dellyhip_split_align(S.ctx, cons.data(), (int32_t)cons.size(), svRefStr.data(), (int32_t)svRefStr.size(), rows.data(), cap, &len,
                                 &found));
  if (!found) return false;   // (`align` is left untouched, src/split.h:494,532)
  // the C-ABI returns _consRefAlignment's orientation (row 0 = consensus); splitAlign itself has the reference first
  align.resize(boost::extents[2][len]);
  for (int32_t j = 0; j < len; ++j) {
    align[0][j] = rows[(std::size_t)cap + j];
    align[1][j] = rows[j];
  }
  return true;
}

/* The loop of src/shortpe.h:175-201 (and :243-268 with sndSeq) as ONE call: for every svid of svidsToProcess
 *   msa(c, seqStore[svid], svs[svid].consensus); alignConsensus(c, hdr, seq, sndSeq, svs[svid])
 * and, when alignConsensus is false, consensus = "", srSupport = 0, srAlignQuality = 0 (:186-190); otherwise srSupport =
 * seqStore[svid].size() (:196).  mapq / srMapQuality (:191-195, host data) stay with the caller; refined[k] tells it which
 * junctions succeeded.  Reads must already be oriented (_adjustOrientation, src/shortpe.h:137) and are taken in the
 * container's ITERATION ORDER.  Returns the number of refined junctions. */
template <typename TConfig, typename TSVs, typename TSeqStore>
inline int refineBatch(TConfig const& c, bam_hdr_t const* hdr, char const* seq, char const* sndSeq, TSVs& svs, TSeqStore const& seqStore,
                       std::vector<uint32_t> const& svidsToProcess, std::vector<uint8_t>* refined = nullptr) {
  namespace dd = dellyhip_dropin;
  if (refined) refined->assign(svidsToProcess.size(), 0);
  if (svidsToProcess.empty()) return 0;
  // compact payload: the exact alleles of small deletions are ~70 % of the result bytes and plain substrings of `seq`, which this
  // caller holds -- they are re-cut here instead of crossing PCIe (store_result below)
  dellyhip_params params = dd::make_params(c, false);
  params.reserved |= DELLYHIP_COMPACT_ALLELES;
  dd::Session& S = dd::session(params);
  std::vector<dellyhip_junction> J;
  J.reserve(svidsToProcess.size());
  std::string blob;
  std::vector<uint64_t> off(1, 0);
  std::size_t longest = 0;
  for (uint32_t svid : svidsToProcess) {
    StructuralVariantRecord const& sv = svs[svid];
    S.chromosome(sv.chr, seq, (int64_t)hdr->target_len[sv.chr]);
    if (sv.chr2 != sv.chr) {
      if (!sndSeq) throw dd::Error(DELLYHIP_E_ARG, "refineBatch: chr2 != chr needs sndSeq (src/shortpe.h:252)");
      S.chromosome(sv.chr2, sndSeq, (int64_t)hdr->target_len[sv.chr2]);
    }
    const uint64_t first = (uint64_t)off.size() - 1;
    dd::pack_reads(seqStore[svid], blob, off);
    for (std::size_t i = first; i + 1 < off.size(); ++i) longest = std::max<std::size_t>(longest, off[i + 1] - off[i]);
    J.push_back(dellyhip_detail::junction_of(sv, (int32_t)svid, first, (int32_t)(off.size() - 1 - first)));
  }
  std::vector<dellyhip_result> R(J.size());
  std::vector<char> out(J.size() * (6 * longest + (std::size_t)std::max(S.params.indelsize, 0) + 1024) + 4096);
  uint64_t used = 0;
  dd::check(dellyhip_refine_batch(S.ctx, (int32_t)J.size(), J.data(), blob.data(), off.data(), (uint64_t)off.size() - 1, R.data(), out.data(),
                                  out.size(), &used, 0));
  int n_ok = 0;
  for (std::size_t k = 0; k < J.size(); ++k) {
    StructuralVariantRecord& sv = svs[svidsToProcess[k]];
    dellyhip_result const& r = R[k];
    if (r.status) throw dd::Error(r.status, "refineBatch: junction beyond a kernel limit (INTEGRATION.md 5)");
    sv.consensus.assign(out.data() + r.cons_off, (std::size_t)std::max(r.cons_len, 0));
    if (!r.ok) {
      sv.consensus = "";
      sv.srSupport = 0;
      sv.srAlignQuality = 0;
      continue;
    }
    dellyhip_detail::store_result(sv, r, out.data(), S.params, J[k], seq, (int64_t)hdr->target_len[sv.chr]);
    sv.srSupport = (int32_t)seqStore[svidsToProcess[k]].size();
    if (refined) (*refined)[k] = 1;
    ++n_ok;
  }
  return n_ok;
}

}  // namespace torali

#endif
