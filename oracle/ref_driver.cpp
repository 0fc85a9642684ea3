// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// oracle/_ref/libdelly_ref.so: the REFERENCE ITSELF on the hot path.  This
// translation unit #includes the reference's own headers from where they lie
// under /root/reference/src (tags.h, edlib.h, msa.h -> align.h gotoh.h needle.h,
// split.h) UNMODIFIED, against the container-only shim in oracle/shim/
// (SURVEY.md 8c), and exposes them through a small C interface that tests,
// the golden-vector generator and bench.py's cpu_baseline leg drive.
//
// Restated here (because src/util.h drags in boost::iostreams/filesystem and
// htslib and cannot be included): reverseComplement (src/util.h:549-563),
// infixStart/infixEnd (src/util.h:86-99), _addAlleles (src/util.h:250-253).
// They keep the reference's observable behaviour, including the quirk that a
// non-ACGTN letter leaves the un-reversed original byte in place.
//
// Build: oracle/Makefile (g++ -std=c++17 -O3 -fno-tree-vectorize -DNDEBUG,
// the reference's release flags, Makefile:47 of the reference).
#include "ref_prelude.h"

#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_set>

#include "edlib.h"
#include "tags.h"

namespace torali {

// src/util.h:86-99
inline uint32_t infixStart(EdlibAlignResult const& cigar) {
  int32_t tIdx = cigar.endLocations[0];
  for (int32_t i = 0; i < cigar.alignmentLength; ++i)
    if (cigar.alignment[i] != EDLIB_EDOP_INSERT) --tIdx;
  return (tIdx >= 0) ? (uint32_t)(tIdx + 1) : 0u;
}
inline uint32_t infixEnd(EdlibAlignResult const& cigar) { return cigar.endLocations[0]; }

// src/util.h:250-253
inline std::string _addAlleles(std::string const& ref, std::string const& alt) {
  return ref + "," + alt;
}

// src/util.h:549-563
inline void reverseComplement(std::string& sequence) {
  std::size_t n = sequence.size();
  std::string up(n, ' ');
  for (std::size_t i = 0; i < n; ++i) up[i] = (char)std::toupper((unsigned char)sequence[n - 1 - i]);
  for (std::size_t i = 0; i < n; ++i) {
    switch (up[i]) {
      case 'A': sequence[i] = 'T'; break;
      case 'C': sequence[i] = 'G'; break;
      case 'G': sequence[i] = 'C'; break;
      case 'T': sequence[i] = 'A'; break;
      case 'N': sequence[i] = 'N'; break;
      default: break;  // byte i keeps its ORIGINAL (un-reversed) value
    }
  }
}

}  // namespace torali

#include "msa.h"
#include "split.h"
#include <numeric>          // std::iota (src/assemble.h:374 relies on a transitive include)
#include "assemble_msa.h"   // src/assemble.h up to assemble(): derived into a temp dir at build time (oracle/Makefile)
#include "coverage_jobs.h"  // src/coverage.h:87-162 (AlignJob, AlignResult, _editDistanceHW, _cutRef*): derived likewise

#include "../include/dellyhip.h"

namespace {

struct RefConfig {  // the duck-typed TConfig of the path
  torali::DnaScore<int> aliscore;
  uint32_t minCliqueSize;
  float flankQuality;
  int32_t minimumFlankSize;
  int32_t indelsize;
  int32_t minConsWindow;
};

RefConfig make_config(const dellyhip_params* p) {
  RefConfig c;
  c.aliscore = torali::DnaScore<int>(p->match, p->mismatch, p->gap_open, p->gap_extend);
  c.minCliqueSize = (uint32_t)p->min_clique_size;
  c.flankQuality = p->flank_quality;
  c.minimumFlankSize = p->minimum_flank_size;
  c.indelsize = p->indelsize;
  c.minConsWindow = p->min_cons_window;
  return c;
}

typedef boost::multi_array<char, 2> TAlign;

void to_align(const char* a, int r, int m, TAlign& out) {
  out.resize(boost::extents[r][m]);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < m; ++j) out[i][j] = a[(size_t)i * m + j];
}

struct BlobWriter {
  char* base;
  uint64_t cap;
  std::atomic<uint64_t> used;
  BlobWriter(char* b, uint64_t c) : base(b), cap(c), used(0) {}
  // returns offset or UINT64_MAX on overflow
  uint64_t put(const char* p, uint64_t n) {
    uint64_t off = used.fetch_add(n);
    if (off + n > cap || base == NULL) return UINT64_MAX;
    if (n) std::memcpy(base + off, p, n);
    return off;
  }
};

void refine_one(RefConfig const& c, bam_hdr_t const* hdr, const char* const* chr_seq,
                const dellyhip_junction& J, const char* blob, const uint64_t* off,
                dellyhip_result& R, BlobWriter& bw, int with_msa, int want_alignment, bool realign) {
  using namespace torali;
  std::memset(&R, 0, sizeof(R));
  R.svid = J.svid;
  R.score_unsplit = R.score_best = R.cons_left = R.ref_left = R.ref_right = -1;  // not observable
  StructuralVariantRecord sv;
  sv.chr = J.chr;
  sv.chr2 = J.chr2;
  sv.svStart = J.sv_start;
  sv.svEnd = J.sv_end;
  sv.svt = J.svt;
  sv.insLen = J.ins_len;
  sv.id = J.svid;
  R.sv_start = J.sv_start;
  R.sv_end = J.sv_end;
  R.ins_len = J.ins_len;

  if (with_msa) {
    // src/shortpe.h:166-171: <=1 read -> no consensus, junction skipped
    if (with_msa != 2 && J.n_seq <= 1) return;
    if (J.n_seq < 1) return;
    std::vector<std::string> sps;  // iteration order == the host's set order
    for (int32_t k = 0; k < J.n_seq; ++k)
      sps.push_back(std::string(blob + off[J.seq_first + k], blob + off[J.seq_first + k + 1]));
    if (with_msa == 2 && sv.svt == 4) {  // src/assemble.h:855-857
      const char* sq = chr_seq[J.chr];
      int32_t seqlen = (int32_t)hdr->target_len[J.chr];
      std::string prefix = boost::to_upper_copy(std::string(sq + std::max(sv.svStart - (int32_t)c.minConsWindow, 0), sq + sv.svStart));
      std::string suffix = boost::to_upper_copy(std::string(sq + sv.svStart, sq + std::min(seqlen, sv.svStart + c.minConsWindow)));
      R.sr_support = msaWfa(c, sps, sv.consensus, prefix, suffix);
      realign = false;                   // :859
    } else if (with_msa == 2) R.sr_support = msaEdlib(c, sps, sv.consensus);  // src/assemble.h:839
    else R.sr_support = msa(c, sps, sv.consensus);  // src/shortpe.h:185
  } else {
    sv.consensus = std::string(blob + off[J.seq_first], blob + off[J.seq_first + 1]);
    R.sr_support = 0;
  }
  std::string consIn = sv.consensus;
  R.cons_len = (int32_t)consIn.size();
  R.cons_off = bw.put(consIn.data(), consIn.size());
  // src/assemble.h:840-848: "take care of small inversions" -- the long-read loop aligns only the
  // middle svSize letters of the consensus and restores it afterwards (:850-853)
  std::string tmpCons;
  int32_t offsetTmpCons = 0;
  if (with_msa == 2 && sv.svt != 4) {
    int32_t svSize = sv.svEnd - sv.svStart;
    if (((sv.svt == 0) || (sv.svt == 1)) && (svSize < (int32_t)sv.consensus.size())) {
      offsetTmpCons = (sv.consensus.size() - svSize) / 2;
      tmpCons = sv.consensus;
      sv.consensus = sv.consensus.substr(offsetTmpCons, svSize);
    }
  }

  // Diagnostics: replay the inner calls of alignConsensus (src/split.h:646-666,
  // :582,:596) with the reference's own functions to expose the alignment rows
  // and the AlignDescriptor, which alignConsensus() itself does not return.
  const char* seq = chr_seq[J.chr];
  const char* sndSeq = (J.chr2 != J.chr) ? chr_seq[J.chr2] : NULL;
  if (!((int32_t)sv.consensus.size() < (2 * c.minimumFlankSize + sv.insLen))) {
    Breakpoint bp(sv);
    if (sv.svt == 4) {
      int32_t bufferSpace = std::max((int32_t)((sv.consensus.size() - sv.insLen) / 3), c.minimumFlankSize);
      _initBreakpoint(hdr, bp, bufferSpace, sv.svt);
    } else _initBreakpoint(hdr, bp, sv.consensus.size(), sv.svt);
    if (bp.chr != bp.chr2) bp.part1 = _getSVRef(c, sndSeq, bp, bp.chr2, sv.svt);
    std::string svRefStr = _getSVRef(c, seq, bp, bp.chr, sv.svt);
    R.ref_len = (int32_t)svRefStr.size();
    std::string consDiag = sv.consensus;
    if (realign) {  // src/split.h:564-572, replayed for the diagnostics below
      std::string revc = consDiag;
      reverseComplement(revc);
      EdlibAlignResult aF = edlibAlign(svRefStr.c_str(), svRefStr.size(), consDiag.c_str(), consDiag.size(), edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0));
      EdlibAlignResult aR = edlibAlign(svRefStr.c_str(), svRefStr.size(), revc.c_str(), revc.size(), edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0));
      if (aR.editDistance < aF.editDistance) consDiag = revc;
      edlibFreeAlignResult(aF);
      edlibFreeAlignResult(aR);
    }
    TAlign align;
    if (_consRefAlignment(consDiag, svRefStr, align, sv.svt)) {
      if (want_alignment) {
        uint64_t len = align.shape()[1];
        std::string rows(2 * len, ' ');
        for (uint64_t j = 0; j < len; ++j) {
          rows[j] = align[0][j];
          rows[len + j] = align[1][j];
        }
        R.aln_off = bw.put(rows.data(), rows.size());
        R.aln_len = (int32_t)len;
      }
      AlignDescriptor ad;
      if (_findSplit(c, consDiag, svRefStr, align, ad, sv.svt)) {
        R.c_start = ad.cStart; R.c_end = ad.cEnd; R.r_start = ad.rStart; R.r_end = ad.rEnd;
        R.hom_left = ad.homLeft; R.hom_right = ad.homRight;
      }
    }
  }

  // The authoritative call: src/shortpe.h:186 / src/split.h:668-672
  bool ok = alignConsensus(c, const_cast<bam_hdr_t const*>(hdr), seq, sndSeq, sv, realign);
  if (!tmpCons.empty()) {  // src/assemble.h:850-853
    sv.consensus = tmpCons;
    sv.consBp += offsetTmpCons;
  }
  if (realign && sv.consensus != consIn && sv.consensus.size() == consIn.size() && R.cons_off != UINT64_MAX)
    std::memcpy(bw.base + R.cons_off, sv.consensus.data(), sv.consensus.size());  // the orientation test replaced sv.consensus
  R.ok = ok ? 1 : 0;
  if (ok) {
    R.sv_start = sv.svStart;
    R.sv_end = sv.svEnd;
    R.ci_wiggle = sv.ciposhigh;
    R.ins_len = sv.insLen;
    R.cons_bp = sv.consBp;
    R.hom_len = sv.homLen;
    R.sr_align_quality = sv.srAlignQuality;
    if (!sv.alleles.empty()) {
      R.allele_off = bw.put(sv.alleles.data(), sv.alleles.size());
      R.allele_len = (int32_t)sv.alleles.size();
    }
  }
  R.matches = R.mismatches = -1;  // not observable through the reference API
}

}  // namespace

extern "C" {

// int lcs(s1,s2)  src/msa.h:10-30
int dref_lcs(const char* a, int la, const char* b, int lb) {
  return torali::lcs(std::string(a, a + la), std::string(b, b + lb));
}

// longestHomology  src/needle.h:13-42
int dref_longest_homology(const char* a, int la, const char* b, int lb, int thr) {
  return torali::longestHomology(std::string(a, a + la), std::string(b, b + lb), thr);
}

// reverseComplement (restated above)
void dref_reverse_complement(char* s, int n) {
  std::string t(s, s + n);
  torali::reverseComplement(t);
  std::memcpy(s, t.data(), n);
}

// longNeedle(cons, ref, align, AlignConfig<true,false>, DnaScore(1,-1,-1,-1))
// src/needle.h:45-222 exactly as called from src/split.h:543-555.
int dref_long_needle(const char* s1, int m, const char* s2, int n, char* rows, int cap, int* len) {
  using namespace torali;
  AlignConfig<true, false> semiglobal;
  DnaScore<int> lnsc(1, -1, -1, -1);
  TAlign aln;
  bool ok = longNeedle(std::string(s1, s1 + m), std::string(s2, s2 + n), aln, semiglobal, lnsc);
  *len = 0;
  if (!ok) return 0;
  int L = (int)aln.shape()[1];
  *len = L;
  if (L > cap) return -1;
  for (int j = 0; j < L; ++j) {
    rows[j] = aln[0][j];
    rows[(size_t)cap + j] = aln[1][j];
  }
  return 1;
}

// edlibAlign(query, target, {k=-1, mode, task, no extra equalities})  src/edlib.cpp:139-300,
// as called by splitAlign (src/split.h:485-527) and _alignConsensus (:568-569).
// mode: 0 NW, 1 SHW, 2 HW (EdlibAlignMode); task: 0 DISTANCE, 1 LOC, 2 PATH.
// out[4] = {editDistance, numLocations, endLocations[0], startLocations[0]} (locations -2 when absent).
// Returns alignmentLength (ops copied to aln when it fits cap).
int dref_edlib_align(const char* q, int qn, const char* t, int tn, int mode, int task, int* out,
                     unsigned char* aln, int cap) {
  // mode | 16: with the 20 extended-IUPAC equality pairs of msaEdlib (src/assemble.h:425)
  EdlibEqualityPair additionalEqualities[20] = {{'M', 'A'}, {'M', 'C'}, {'R', 'A'}, {'R', 'G'}, {'W', 'A'}, {'W', 'T'}, {'B', 'A'}, {'B', '-'}, {'S', 'C'}, {'S', 'G'}, {'Y', 'C'}, {'Y', 'T'}, {'D', 'C'}, {'D', '-'}, {'K', 'G'}, {'K', 'T'}, {'E', 'G'}, {'E', '-'}, {'F', 'T'}, {'F', '-'}};
  const bool iupac = (mode & 16) != 0;
  mode &= 15;
  EdlibAlignResult r = edlibAlign(q, qn, t, tn, edlibNewAlignConfig(-1, (EdlibAlignMode)mode, (EdlibAlignTask)task, iupac ? additionalEqualities : NULL, iupac ? 20 : 0));
  out[0] = r.editDistance;
  out[1] = r.numLocations;
  out[2] = r.endLocations ? r.endLocations[0] : -2;
  out[3] = r.startLocations ? r.startLocations[0] : -2;
  int L = r.alignmentLength;
  if (r.alignment && L <= cap) std::memcpy(aln, r.alignment, (size_t)L);
  edlibFreeAlignResult(r);
  return L;
}

// edlibAlign with k and every field of EdlibAlignResult (all locations), for the parity tests of dellyhip_edlib_align_full
// and include/delly_dropin/edlib.h.  out[5] = {status, editDistance, numLocations, alignmentLength, alphabetLength};
// ends / starts: up to cap entries (starts[0] = -2 when the reference returns no start locations).
int dref_edlib_align_full(const char* q, int qn, const char* t, int tn, int k, int mode, int task, int iupac, int* out,
                          int* ends, int* starts, int cap, unsigned char* aln, int acap) {
  EdlibEqualityPair additionalEqualities[20] = {{'M', 'A'}, {'M', 'C'}, {'R', 'A'}, {'R', 'G'}, {'W', 'A'}, {'W', 'T'}, {'B', 'A'}, {'B', '-'}, {'S', 'C'}, {'S', 'G'}, {'Y', 'C'}, {'Y', 'T'}, {'D', 'C'}, {'D', '-'}, {'K', 'G'}, {'K', 'T'}, {'E', 'G'}, {'E', '-'}, {'F', 'T'}, {'F', '-'}};
  EdlibAlignResult r = edlibAlign(q, qn, t, tn, edlibNewAlignConfig(k, (EdlibAlignMode)mode, (EdlibAlignTask)task, iupac ? additionalEqualities : NULL, iupac ? 20 : 0));
  out[0] = r.status;
  out[1] = r.editDistance;
  out[2] = r.numLocations;
  out[3] = r.alignmentLength;
  out[4] = r.alphabetLength;
  if (cap > 0) starts[0] = -2;
  for (int i = 0; i < r.numLocations && i < cap; ++i) {
    if (r.endLocations) ends[i] = r.endLocations[i];
    if (r.startLocations) starts[i] = r.startLocations[i];
  }
  if (r.alignment && r.alignmentLength <= acap) std::memcpy(aln, r.alignment, (size_t)r.alignmentLength);
  edlibFreeAlignResult(r);
  return 0;
}

// edlibAlignmentToCigar (src/edlib.cpp:294-343): returns the length of the text (copied when it fits), -1 for NULL
int dref_edlib_cigar(const unsigned char* aln, int n, int fmt, char* out, int cap) {
  char* c = edlibAlignmentToCigar(aln, n, (EdlibCigarFormat)fmt);
  if (!c) return -1;
  const int len = (int)std::strlen(c);
  if (len < cap) std::memcpy(out, c, (size_t)len + 1);
  free(c);
  return len;
}

// splitAlign(cons, svRefStr, align)  src/split.h:480-538 followed by the row swap of
// _consRefAlignment (:546-552).  Returns 1/0, rows[0..len) = consensus row, rows[cap..cap+len) = ref row.
int dref_split_align(const char* cons, int m, const char* ref, int n, char* rows, int cap, int* len) {
  using namespace torali;
  TAlign aln;
  bool ok = _consRefAlignment(std::string(cons, cons + m), std::string(ref, ref + n), aln, 4);
  int L = (int)aln.shape()[1];
  *len = L;
  if (L > cap) return -1;
  for (int j = 0; j < L; ++j) {
    rows[j] = aln[0][j];
    rows[(size_t)cap + j] = aln[1][j];
  }
  return ok ? 1 : 0;
}

// gotoh(a1, a2, align, AlignConfig<true,true>, c.aliscore)  src/gotoh.h:71-174
// as called by palign, src/msa.h:106-107.
int dref_gotoh(const dellyhip_params* p, const char* a1, int r1, int m, const char* a2, int r2,
               int n, char* out, int cap, int* len) {
  using namespace torali;
  RefConfig c = make_config(p);
  TAlign A1, A2, A;
  to_align(a1, r1, m, A1);
  to_align(a2, r2, n, A2);
  AlignConfig<true, true> endFree;
  int score = gotoh(A1, A2, A, endFree, c.aliscore);
  int L = (int)A.shape()[1];
  *len = L;
  if (L <= cap)
    for (int i = 0; i < r1 + r2; ++i)
      for (int j = 0; j < L; ++j) out[(size_t)i * cap + j] = A[i][j];
  return score;
}

// consensus(c, align, cs)  src/msa.h:175-183
int dref_consensus(const dellyhip_params* p, const char* a, int r, int m, char* cs, int cap) {
  using namespace torali;
  RefConfig c = make_config(p);
  TAlign A;
  to_align(a, r, m, A);
  std::string s;
  consensus(c, A, s);
  if ((int)s.size() <= cap) std::memcpy(cs, s.data(), s.size());
  return (int)s.size();
}

// msa(c, sps, cs)  src/msa.h:185-239; returns rows, consensus in cs
int dref_msa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs,
             int cap, int* cs_len) {
  using namespace torali;
  RefConfig c = make_config(p);
  std::vector<std::string> sps;
  for (int k = 0; k < n_reads; ++k) sps.push_back(std::string(blob + off[k], blob + off[k + 1]));
  std::string s;
  int rows = msa(c, sps, s);
  *cs_len = (int)s.size();
  if ((int)s.size() <= cap) std::memcpy(cs, s.data(), s.size());
  return rows;
}

// msaEdlib(c, sps, cs)  src/assemble.h:383-473; returns rows, consensus in cs
int dref_msa_edlib(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs,
                   int cap, int* cs_len) {
  using namespace torali;
  RefConfig c = make_config(p);
  std::vector<std::string> sps;
  for (int k = 0; k < n_reads; ++k) sps.push_back(std::string(blob + off[k], blob + off[k + 1]));
  std::string s;
  int rows = msaEdlib(c, sps, s);
  *cs_len = (int)s.size();
  if ((int)s.size() <= cap) std::memcpy(cs, s.data(), s.size());
  return rows;
}

// msaWfa(c, sps, cs, prefix, suffix)  src/assemble.h:547-726; returns rows, consensus in cs
int dref_msa_wfa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, const char* prefix, int pn,
                 const char* suffix, int sn, char* cs, int cap, int* cs_len) {
  using namespace torali;
  RefConfig c = make_config(p);
  std::vector<std::string> sps;
  for (int k = 0; k < n_reads; ++k) sps.push_back(std::string(blob + off[k], blob + off[k + 1]));
  std::string s;
  int rows = msaWfa(c, sps, s, std::string(prefix, prefix + pn), std::string(suffix, suffix + sn));
  *cs_len = (int)s.size();
  if ((int)s.size() <= cap) std::memcpy(cs, s.data(), s.size());
  return rows;
}

// distanceMatrix + upgma (src/msa.h:32-89): returns root, fills d[(2n+1)^2], p[(2n+1)*3]
int dref_guide_tree(int n_reads, const char* blob, const uint64_t* off, int* dflat, int* pflat) {
  using namespace torali;
  typedef boost::multi_array<int, 2> TDist;
  typedef TDist::index TDIndex;
  std::vector<std::string> sps;
  for (int k = 0; k < n_reads; ++k) sps.push_back(std::string(blob + off[k], blob + off[k + 1]));
  TDIndex num = n_reads;
  TDist d(boost::extents[2 * num + 1][2 * num + 1]);
  for (TDIndex i = 0; i < (2 * num + 1); ++i)
    for (TDIndex j = i + 1; j < (2 * num + 1); ++j) d[i][j] = -1;
  distanceMatrix(sps, d);
  if (dflat)
    for (TDIndex i = 0; i < (2 * num + 1); ++i)
      for (TDIndex j = 0; j < (2 * num + 1); ++j) dflat[i * (2 * num + 1) + j] = d[i][j];
  TDist ph(boost::extents[2 * num + 1][3]);
  for (TDIndex i = 0; i < (2 * num + 1); ++i)
    for (TDIndex j = 0; j < 3; ++j) ph[i][j] = -1;
  TDIndex root = upgma(d, ph, num);
  for (TDIndex i = 0; i < (2 * num + 1); ++i)
    for (TDIndex j = 0; j < 3; ++j) pflat[i * 3 + j] = ph[i][j];
  return (int)root;
}

// The loop body of src/shortpe.h:175-201 over a batch, with the reference's
// threading model: n_threads std::threads pulling junction indices from one
// std::atomic counter.  with_msa=0: alignConsensus only (BASELINE unit U).
int dref_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                      const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                      const char* blob, const uint64_t* off, dellyhip_result* results,
                      char* out_blob, uint64_t out_cap, uint64_t* out_used, int with_msa,
                      int want_alignment, int n_threads) {
  RefConfig c = make_config(p);
  std::vector<uint32_t> tlen(n_chr);
  for (int i = 0; i < n_chr; ++i) tlen[i] = (uint32_t)chr_len[i];
  bam_hdr_t hdr;
  hdr.n_targets = n_chr;
  hdr.target_len = tlen.data();
  hdr.target_name = NULL;
  BlobWriter bw(out_blob, out_cap);
  std::atomic<uint32_t> next(0);
  auto worker = [&]() {
    for (;;) {
      uint32_t idx = next.fetch_add(1, std::memory_order_relaxed);
      if (idx >= (uint32_t)n_junc) break;
      refine_one(c, &hdr, chr_seq, junc[idx], blob, off, results[idx], bw, with_msa, want_alignment, (p->reserved & 1) != 0);
    }
  };
  if (n_threads <= 1) worker();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (out_used) *out_used = bw.used.load();
  return (bw.used.load() > out_cap && out_blob) ? -1 : 0;
}

// CPU baseline timer (bench.py cpu_baseline): ONLY the loop body of src/shortpe.h:183-197 /
// src/assemble.h:836-861 -- msa()/msaEdlib()/msaWfa() when reads are given, then ONE alignConsensus() --
// no diagnostic replay, no blob marshalling.  n_threads std::threads pull indices from one atomic counter
// (the reference's model, src/shortpe.h:175-182) until reps * n_junc junction visits are done; the clock
// (steady_clock, inside this function) covers thread start to join.  *seconds = wall time of that region,
// *n_ok = alignConsensus() calls that returned true (keeps the work observable).
int dref_time_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                           const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                           const char* blob, const uint64_t* off, int with_msa, int n_threads, int reps,
                           double* seconds, int64_t* n_ok) {
  using namespace torali;
  RefConfig c = make_config(p);
  std::vector<uint32_t> tlen(n_chr);
  for (int i = 0; i < n_chr; ++i) tlen[i] = (uint32_t)chr_len[i];
  bam_hdr_t hdr;
  hdr.n_targets = n_chr;
  hdr.target_len = tlen.data();
  hdr.target_name = NULL;
  const bool realign0 = (p->reserved & 1) != 0;
  const uint64_t total = (uint64_t)std::max(reps, 1) * (uint64_t)std::max(n_junc, 0);
  std::atomic<uint64_t> next(0);
  std::atomic<int64_t> oks(0);
  auto worker = [&]() {
    int64_t mine = 0;
    for (;;) {
      const uint64_t t = next.fetch_add(1, std::memory_order_relaxed);
      if (t >= total) break;
      const dellyhip_junction& J = junc[t % (uint64_t)n_junc];
      StructuralVariantRecord sv;
      sv.chr = J.chr; sv.chr2 = J.chr2; sv.svStart = J.sv_start; sv.svEnd = J.sv_end; sv.svt = J.svt;
      sv.insLen = J.ins_len; sv.id = J.svid;
      bool realign = realign0;
      if (with_msa) {
        if (with_msa != 2 && J.n_seq <= 1) continue;
        if (J.n_seq < 1) continue;
        std::vector<std::string> sps;
        for (int32_t k = 0; k < J.n_seq; ++k)
          sps.push_back(std::string(blob + off[J.seq_first + k], blob + off[J.seq_first + k + 1]));
        if (with_msa == 2 && sv.svt == 4) {
          const char* sq = chr_seq[J.chr];
          int32_t seqlen = (int32_t)hdr.target_len[J.chr];
          std::string prefix = boost::to_upper_copy(std::string(sq + std::max(sv.svStart - (int32_t)c.minConsWindow, 0), sq + sv.svStart));
          std::string suffix = boost::to_upper_copy(std::string(sq + sv.svStart, sq + std::min(seqlen, sv.svStart + c.minConsWindow)));
          msaWfa(c, sps, sv.consensus, prefix, suffix);
          realign = false;
        } else if (with_msa == 2) msaEdlib(c, sps, sv.consensus);
        else msa(c, sps, sv.consensus);
      } else sv.consensus = std::string(blob + off[J.seq_first], blob + off[J.seq_first + 1]);
      if (with_msa == 2 && sv.svt != 4) {   // src/assemble.h:840-848
        int32_t svSize = sv.svEnd - sv.svStart;
        if (((sv.svt == 0) || (sv.svt == 1)) && (svSize < (int32_t)sv.consensus.size()))
          sv.consensus = sv.consensus.substr((sv.consensus.size() - svSize) / 2, svSize);
      }
      const char* seq = chr_seq[J.chr];
      const char* sndSeq = (J.chr2 != J.chr) ? chr_seq[J.chr2] : NULL;
      if (alignConsensus(c, const_cast<bam_hdr_t const*>(&hdr), seq, sndSeq, sv, realign)) ++mine;
    }
    oks.fetch_add(mine);
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (n_threads <= 1) worker();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (n_ok) *n_ok = oks.load();
  return 0;
}

// std::unordered_set<std::string> iteration order of the host
// (src/shortpe.h:68,96): inserts the reads in the given order and returns the
// permutation in which the set iterates (perm[k] = input index, -1 padded when
// duplicates collapse).  Returns the set size.
int dref_unordered_set_order(int n_reads, const char* blob, const uint64_t* off, int* perm) {
  std::unordered_set<std::string> s;
  std::vector<std::string> in;
  for (int k = 0; k < n_reads; ++k) {
    in.push_back(std::string(blob + off[k], blob + off[k + 1]));
    s.insert(in.back());
  }
  int k = 0;
  for (auto const& x : s) {
    int idx = -1;
    for (int i = 0; i < n_reads; ++i)
      if (in[i] == x) { idx = i; break; }
    perm[k++] = idx;
  }
  for (int i = k; i < n_reads; ++i) perm[i] = -1;
  return (int)s.size();
}

}  // extern "C"


// ---- split-read genotyping classifier: the worker body of process_batch, src/coverage.h:418-434 ----
// _editDistanceHW, AlignJob and AlignResult are the reference's (coverage_jobs.h); the ten lines of the
// lambda body live inside genotype code that needs htslib, so they are replayed here statement by statement.
extern "C" int dref_classify_reads(const dellyhip_params* p, uint64_t n_jobs, const dellyhip_align_job* jobs,
                                   const char* blob, dellyhip_align_result* out, int n_threads, int with_dist,
                                   double* worker_seconds) {
  RefConfig c = make_config(p);
  // the job buffer of src/coverage.h:411,541 (strings are copied when the BAM loop pushes a job, not by the workers)
  std::vector<torali::AlignJob> jobBuf;
  jobBuf.reserve(n_jobs);
  for (uint64_t i = 0; i < n_jobs; ++i) {
    const dellyhip_align_job& J = jobs[i];
    jobBuf.push_back(torali::AlignJob(std::string(blob + J.cons_off, J.cons_len), std::string(blob + J.ref_off, J.ref_len),
                                      std::string(blob + J.seq_off, J.seq_len), J.file_index, J.sv_id, J.qual));
  }
  std::vector<torali::AlignResult> results(n_jobs, torali::AlignResult());
  std::atomic<uint64_t> next(0);   // as process_batch: workers pull job indices from one atomic counter (:414-417)
  auto worker = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_jobs) break;
      torali::AlignJob& job = jobBuf[i];
      double scoreAlt = torali::_editDistanceHW(c, job.consProbe, job.sequence);
      double scoreRef = torali::_editDistanceHW(c, job.refProbe, job.sequence);
      if ((scoreRef > 0.7) || (scoreAlt > 0.7)) {
        results[i].svId = job.svId;
        results[i].fileIndex = job.fileIndex;
        if (scoreRef > scoreAlt) {
          results[i].type = 'R';
          results[i].qual = (uint8_t) std::min(255, std::min((int) (scoreRef * 35), (int) job.qual));
        } else {
          results[i].type = 'A';
          results[i].qual = (uint8_t) std::min(255, std::min((int) (scoreAlt * 35), (int) job.qual));
        }
      }
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (n_threads <= 1) worker();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (worker_seconds) *worker_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (uint64_t i = 0; i < n_jobs; ++i) {
    out[i].file_index = results[i].fileIndex;
    out[i].sv_id = results[i].svId;
    out[i].type = (uint8_t)results[i].type;
    out[i].qual = results[i].qual;
    out[i].status = 0;
    out[i].dist_alt = out[i].dist_ref = 0;
    if (with_dist) {   // the two distances behind the scores, from the reference's own edlib with the reference's k
      auto dist = [&](std::string const& q) {
        EdlibAlignResult a = edlibAlign(q.c_str(), q.size(), jobBuf[i].sequence.c_str(), jobBuf[i].sequence.size(),
                                        edlibNewAlignConfig(2 * c.flankQuality * q.size(), EDLIB_MODE_HW, EDLIB_TASK_DISTANCE, NULL, 0));
        int d = a.editDistance;
        edlibFreeAlignResult(a);
        return d;
      };
      out[i].dist_alt = dist(jobBuf[i].consProbe);
      out[i].dist_ref = dist(jobBuf[i].refProbe);
    }
  }
  return 0;
}


// ---- long-read genotyping: the reference's _editDistanceNW (src/genotype.h:21-30, derived header) per pair ----
extern "C" int dref_edit_distance_nw_batch(uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob, int32_t* out,
                                           int n_threads, double* worker_seconds) {
  std::vector<std::string> q(n_jobs), t(n_jobs);   // the ref / alt / probe strings exist before the calls (src/genotype.h:270-272)
  for (uint64_t i = 0; i < n_jobs; ++i) {
    q[i].assign(blob + jobs[i].query_off, jobs[i].query_len);
    t[i].assign(blob + jobs[i].target_off, jobs[i].target_len);
  }
  std::atomic<uint64_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_jobs) break;
      out[i] = torali::_editDistanceNW(q[i], t[i]);
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (n_threads <= 1) worker();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (worker_seconds) *worker_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}


// ---- probe generation: the per-SV body of _generateProbes, src/coverage.h:196-258, replayed with the reference's
// own _initBreakpoint / _getSVRef / _consRefAlignment / _findSplit / _cutRefStart / _cutRefEnd (the enclosing
// function needs faidx / htslib, so its statements are repeated here in their order) ----
extern "C" int dref_generate_probes(const dellyhip_params* p, int n_chr, const char* const* chr_seq, const int64_t* chr_len,
                                    int n_junc, const dellyhip_junction* junc, const char* blob, const uint64_t* off,
                                    dellyhip_probes* probes, char* out_blob, uint64_t out_cap, uint64_t* out_used) {
  using namespace torali;
  RefConfig c = make_config(p);
  std::vector<uint32_t> tlen(n_chr);
  for (int i = 0; i < n_chr; ++i) tlen[i] = (uint32_t)chr_len[i];
  bam_hdr_t hdrv;
  hdrv.n_targets = n_chr;
  hdrv.target_len = tlen.data();
  hdrv.target_name = NULL;
  bam_hdr_t* hdr = &hdrv;
  BlobWriter bw(out_blob, out_cap);
  for (int i = 0; i < n_junc; ++i) {
    const dellyhip_junction& J = junc[i];
    dellyhip_probes& O = probes[i];
    std::memset(&O, 0, sizeof(O));
    O.svid = J.svid;
    StructuralVariantRecord svr;
    svr.chr = J.chr; svr.chr2 = J.chr2; svr.svStart = J.sv_start; svr.svEnd = J.sv_end; svr.svt = J.svt;
    svr.insLen = J.ins_len; svr.id = J.svid;
    svr.consensus = std::string(blob + off[J.seq_first], blob + off[J.seq_first + 1]);
    StructuralVariantRecord* itSV = &svr;
    // regions (:233-253) do not depend on the alignment
    for (unsigned int bpPoint = 0; bpPoint < 2; ++bpPoint) {
      if (bpPoint) {
        O.region_start[1] = std::max(0, itSV->svEnd - c.minimumFlankSize);
        O.region_end[1] = std::min((uint32_t) (itSV->svEnd + c.minimumFlankSize), hdr->target_len[itSV->chr2]);
        O.bppos[1] = itSV->svEnd;
      } else {
        O.region_start[0] = std::max(0, itSV->svStart - c.minimumFlankSize);
        O.region_end[0] = std::min((uint32_t) (itSV->svStart + c.minimumFlankSize), hdr->target_len[itSV->chr]);
        O.bppos[0] = itSV->svStart;
      }
    }
    // :196-217
    std::string part1;
    if (itSV->chr != itSV->chr2) {
      Breakpoint bp(*itSV);
      _initBreakpoint(hdr, bp, (int32_t) itSV->consensus.size(), itSV->svt);
      part1 = _getSVRef(c, chr_seq[itSV->chr2], bp, itSV->chr2, itSV->svt);
    }
    Breakpoint bp(*itSV);
    if (_translocation(itSV->svt)) bp.part1 = part1;
    if (itSV->svt == 4) {
      int32_t bufferSpace = std::max((int32_t) ((itSV->consensus.size() - itSV->insLen) / 3), c.minimumFlankSize);
      _initBreakpoint(hdr, bp, bufferSpace, itSV->svt);
    } else _initBreakpoint(hdr, bp, (int32_t) itSV->consensus.size(), itSV->svt);
    std::string svRefStr = _getSVRef(c, chr_seq[itSV->chr], bp, itSV->chr, itSV->svt);
    if (itSV->svt == 4 && svRefStr.size() < 3) { O.status = DELLYHIP_E_LIMIT; continue; }   // splitAlign indexes out of bounds there
    TAlign align;
    if (!_consRefAlignment(itSV->consensus, svRefStr, align, itSV->svt)) continue;
    AlignDescriptor ad;
    if (!_findSplit(c, itSV->consensus, svRefStr, align, ad, itSV->svt)) continue;
    O.ok = 1;
    O.hom_left = ad.homLeft;
    O.hom_right = ad.homRight;
    for (unsigned int bpPoint = 0; bpPoint < 2; ++bpPoint) {   // :230-256
      int32_t cutConsStart, cutConsEnd, cutRefStart, cutRefEnd;
      if (bpPoint) {
        cutConsStart = ad.cEnd - ad.homLeft - c.minimumFlankSize;
        cutConsEnd = ad.cEnd + ad.homRight + c.minimumFlankSize;
      } else {
        cutConsStart = ad.cStart - ad.homLeft - c.minimumFlankSize;
        cutConsEnd = ad.cStart + ad.homRight + c.minimumFlankSize;
      }
      cutRefStart = _cutRefStart(ad.rStart, ad.rEnd, ad.homLeft + c.minimumFlankSize, bpPoint, itSV->svt);
      cutRefEnd = _cutRefEnd(ad.rStart, ad.rEnd, ad.homRight + c.minimumFlankSize, bpPoint, itSV->svt);
      std::string consProbe = itSV->consensus.substr(cutConsStart, (cutConsEnd - cutConsStart));
      std::string refProbe = svRefStr.substr(cutRefStart, (cutRefEnd - cutRefStart));
      O.cons_len[bpPoint] = (int32_t)consProbe.size();
      O.ref_len[bpPoint] = (int32_t)refProbe.size();
      O.cons_off[bpPoint] = bw.put(consProbe.data(), consProbe.size());
      O.ref_off[bpPoint] = bw.put(refProbe.data(), refProbe.size());
    }
  }
  if (out_used) *out_used = bw.used.load();
  return (bw.used.load() > out_cap && out_blob) ? -1 : 0;
}
