// TEST ONLY -- compiled C++ caller of the drop-in headers (include/delly_dropin/*.h).
//
// The translation unit looks like a Delly source file: it includes the reference's own tags.h / align.h (from
// $(REF)/src, against the container shim oracle/shim for Boost / htslib PODs) and then the DROP-IN msa.h, needle.h,
// gotoh.h, split.h, assemble_msa.h instead of the reference's.  It drives the torali:: entry points with the
// reference's call shapes on a batch file written by tests/test_gpu_dropin_cpp.py and writes what the reference's
// caller would observe (StructuralVariantRecord fields, consensus, alleles) to a result file, which the Python side
// compares with oracle/_ref.  No oracle code is linked: the only library is libdellyhip.so.
//
//   dropin_test abi                      -> prints dellyhip_abi_info, exit 0 (no GPU needed)
//   dropin_test run  <in.bin> <out.bin>  -> needs an MI355X
//   dropin_test nodevice                 -> exit 0 iff a call without a usable device FAILS LOUDLY (no CPU path)
#include "ref_prelude.h"

#include <cstdio>
#include <fstream>
#include <set>
#include <unordered_set>

#include "tags.h"

#include "msa.h"            // include/delly_dropin/msa.h          (NOT the reference's)
#include "needle.h"         // include/delly_dropin/needle.h
#include "gotoh.h"          // include/delly_dropin/gotoh.h
#include "split.h"          // include/delly_dropin/split.h
#include "assemble_msa.h"   // include/delly_dropin/assemble_msa.h

namespace {

struct TestConfig {   // the duck-typed TConfig of `delly sr` (src/delly.h:49-82, the fields the path reads)
  torali::DnaScore<int> aliscore;
  uint32_t minCliqueSize;
  float flankQuality;
  int32_t minimumFlankSize;
  int32_t indelsize;
  int32_t minConsWindow;
};

TestConfig config_from(dellyhip_params const& p) {
  TestConfig c;
  c.aliscore = torali::DnaScore<int>(p.match, p.mismatch, p.gap_open, p.gap_extend);
  c.minCliqueSize = (uint32_t)p.min_clique_size;
  c.flankQuality = p.flank_quality;
  c.minimumFlankSize = p.minimum_flank_size;
  c.indelsize = p.indelsize;
  c.minConsWindow = p.min_cons_window;
  return c;
}

template <typename T>
bool rd(std::ifstream& f, T& v) { return (bool)f.read(reinterpret_cast<char*>(&v), sizeof v); }
template <typename T>
void wr(std::ofstream& f, T const& v) { f.write(reinterpret_cast<const char*>(&v), sizeof v); }
void wrs(std::ofstream& f, std::string const& s) {
  uint32_t n = (uint32_t)s.size();
  wr(f, n);
  f.write(s.data(), n);
}

// an insertion-ordered "set": iteration order = the order the Python side packed the reads in
typedef std::vector<std::string> ReadSet;

void write_sv(std::ofstream& out, bool ok, int rows, torali::StructuralVariantRecord const& sv) {
  wr(out, (int32_t)(ok ? 1 : 0));
  wr(out, (int32_t)rows);
  wr(out, sv.svStart); wr(out, sv.svEnd); wr(out, sv.ciposlow); wr(out, sv.ciposhigh); wr(out, sv.ciendlow); wr(out, sv.ciendhigh);
  wr(out, sv.insLen); wr(out, sv.consBp); wr(out, sv.homLen); wr(out, sv.srSupport);
  wr(out, sv.srAlignQuality);
  wr(out, (int32_t)(sv.precise ? 1 : 0));
  wrs(out, sv.consensus);
  wrs(out, sv.alleles);
}

int run(const char* in_path, const char* out_path) {
  std::ifstream in(in_path, std::ios::binary);
  if (!in) { std::fprintf(stderr, "cannot open %s\n", in_path); return 2; }
  dellyhip_params P;
  int32_t realign = 0, mode = 0, n_chr = 0;
  rd(in, P); rd(in, realign); rd(in, mode); rd(in, n_chr);   // mode 0: alignConsensus only, 1: msa, 2: msaEdlib / msaWfa (long reads)
  std::vector<std::string> chroms(n_chr);
  std::vector<uint32_t> tlen(n_chr);
  for (int i = 0; i < n_chr; ++i) {
    uint64_t len = 0;
    rd(in, len);
    chroms[i].resize(len);
    in.read(&chroms[i][0], (std::streamsize)len);
    tlen[i] = (uint32_t)len;
  }
  bam_hdr_t hdr;
  hdr.n_targets = n_chr;
  hdr.target_len = tlen.data();
  hdr.target_name = NULL;
  int32_t n = 0;
  rd(in, n);
  std::vector<dellyhip_junction> J(n);
  if (n) in.read(reinterpret_cast<char*>(J.data()), (std::streamsize)(n * sizeof(dellyhip_junction)));
  uint64_t n_seq = 0;
  rd(in, n_seq);
  std::vector<uint64_t> off(n_seq + 1);
  in.read(reinterpret_cast<char*>(off.data()), (std::streamsize)((n_seq + 1) * 8));
  std::string blob(off[n_seq], ' ');
  in.read(&blob[0], (std::streamsize)blob.size());
  if (!in) { std::fprintf(stderr, "short read of %s\n", in_path); return 2; }

  const TestConfig c = config_from(P);
  std::ofstream out(out_path, std::ios::binary);
  wr(out, n);

  // ---- pass 1: the per-junction calls of src/shortpe.h:185-190 / src/assemble.h:839-860, one by one
  std::vector<torali::StructuralVariantRecord> svs(n);
  std::vector<ReadSet> seqStore(n);
  for (int k = 0; k < n; ++k) {
    torali::StructuralVariantRecord& sv = svs[k];
    sv.chr = J[k].chr; sv.chr2 = J[k].chr2; sv.svStart = J[k].sv_start; sv.svEnd = J[k].sv_end;
    sv.svt = J[k].svt; sv.insLen = J[k].ins_len; sv.id = J[k].svid;
    for (int32_t q = 0; q < J[k].n_seq; ++q)
      seqStore[k].push_back(blob.substr(off[J[k].seq_first + q], off[J[k].seq_first + q + 1] - off[J[k].seq_first + q]));
  }
  for (int k = 0; k < n; ++k) {
    torali::StructuralVariantRecord sv = svs[k];
    int rows = 0;
    bool rl = realign != 0;
    if (mode == 1) rows = torali::msa(c, seqStore[k], sv.consensus);
    else if (mode == 2 && sv.svt == 4) {   // src/assemble.h:855-859
      const std::string& sq = chroms[sv.chr];
      std::string prefix = sq.substr(std::max(sv.svStart - c.minConsWindow, 0), sv.svStart - std::max(sv.svStart - c.minConsWindow, 0));
      std::string suffix = sq.substr(sv.svStart, std::min<int64_t>((int64_t)sq.size(), (int64_t)sv.svStart + c.minConsWindow) - sv.svStart);
      for (auto& ch : prefix) ch = (char)std::toupper((unsigned char)ch);
      for (auto& ch : suffix) ch = (char)std::toupper((unsigned char)ch);
      rows = torali::msaWfa(c, seqStore[k], sv.consensus, prefix, suffix);
      rl = false;
    } else if (mode == 2) rows = torali::msaEdlib(c, seqStore[k], sv.consensus);
    else sv.consensus = seqStore[k][0];
    const char* seq = chroms[sv.chr].data();
    const char* snd = (sv.chr2 != sv.chr) ? chroms[sv.chr2].data() : NULL;
    const bool ok = torali::alignConsensus(c, &hdr, seq, snd, sv, rl);
    write_sv(out, ok, rows, sv);
  }

  // ---- pass 2 (short-read msa mode): the same junctions through torali::refineBatch, one call per chromosome pair
  int32_t n_batch = 0;
  if (mode == 1) {
    std::vector<uint32_t> ids;
    for (int k = 0; k < n; ++k)
      if (seqStore[k].size() > 1) ids.push_back((uint32_t)k);   // src/shortpe.h:166-171
    n_batch = (int32_t)ids.size();
    wr(out, n_batch);
    std::set<std::pair<int, int> > pairs;
    for (uint32_t k : ids) pairs.insert(std::make_pair(svs[k].chr, svs[k].chr2));
    for (auto const& pr : pairs) {
      std::vector<uint32_t> sel;
      for (uint32_t k : ids)
        if (svs[k].chr == pr.first && svs[k].chr2 == pr.second) sel.push_back(k);
      std::vector<uint8_t> fine;
      torali::refineBatch(c, &hdr, chroms[pr.first].data(), (pr.second != pr.first) ? chroms[pr.second].data() : NULL, svs, seqStore, sel, &fine);
      for (std::size_t q = 0; q < sel.size(); ++q) {
        wr(out, (int32_t)sel[q]);
        write_sv(out, fine[q] != 0, (int)seqStore[sel[q]].size(), svs[sel[q]]);
      }
    }
  } else wr(out, n_batch);

  // ---- pass 3: the primitive signatures once each (longNeedle, splitAlign, gotoh, _adjustOrientation)
  {
    typedef boost::multi_array<char, 2> TAlign;
    torali::AlignConfig<true, false> semiglobal;
    torali::DnaScore<int> lnsc(1, -1, -1, -1);
    const std::string ref = chroms[0].substr(1200, 900);
    std::string up(ref);
    for (auto& ch : up) ch = (char)std::toupper((unsigned char)ch);
    const std::string cons = up.substr(100, 80) + up.substr(600, 90);
    TAlign aln;
    const bool f1 = torali::longNeedle(cons, up, aln, semiglobal, lnsc);
    wr(out, (int32_t)f1);
    std::string r0, r1;
    if (f1) for (std::size_t j = 0; j < aln.shape()[1]; ++j) { r0 += aln[0][j]; r1 += aln[1][j]; }
    wrs(out, cons); wrs(out, up); wrs(out, r0); wrs(out, r1);
    const std::string insCons = up.substr(300, 90) + std::string("ACGTTGCATTGACCAGTACCATGGATCAGTTTGACACAGT") + up.substr(390, 90);
    const std::string insRef = up.substr(330, 120);   // the window _initBreakpoint cuts for this insertion: (220 - 40) / 3 = 60 bp per side
    TAlign sa;
    const bool f2 = torali::splitAlign(insCons, insRef, sa);
    wr(out, (int32_t)f2);
    std::string s0, s1;
    if (f2) for (std::size_t j = 0; j < sa.shape()[1]; ++j) { s0 += sa[0][j]; s1 += sa[1][j]; }
    wrs(out, insCons); wrs(out, insRef); wrs(out, s0); wrs(out, s1);
    TAlign a1(boost::extents[1][60]), a2(boost::extents[1][70]), ga;
    for (int j = 0; j < 60; ++j) a1[0][j] = up[500 + j];
    for (int j = 0; j < 70; ++j) a2[0][j] = up[495 + j];
    torali::AlignConfig<true, true> endFree;
    const int sc = torali::gotoh(a1, a2, ga, endFree, c.aliscore);
    wr(out, (int32_t)sc);
    wr(out, (int32_t)ga.shape()[0]);
    for (std::size_t i = 0; i < ga.shape()[0]; ++i) {
      std::string row;
      for (std::size_t j = 0; j < ga.shape()[1]; ++j) row += ga[i][j];
      wrs(out, row);
    }
    std::string a = "ACGTNacgtRYKM", b = a, d = a;
    torali::_adjustOrientation(a, true, 0);    // inversion 3to3, far side: flipped
    torali::_adjustOrientation(b, true, 1);    // inversion 5to5, far side: kept
    torali::_adjustOrientation(d, false, 6);   // translocation 5to5, near side: flipped
    wrs(out, a); wrs(out, b); wrs(out, d);
  }
  // ---- pass 4: msa() on the host's own container, std::unordered_set<std::string> (src/shortpe.h:68):
  // duplicates collapse in the set and the reads reach the device in the set's ITERATION order
  {
    const int nset = (mode == 1) ? std::min(n, 6) : 0;
    wr(out, (int32_t)nset);
    for (int k = 0; k < nset; ++k) {
      std::unordered_set<std::string> us;
      for (auto const& r : seqStore[k]) us.insert(r);
      if (!seqStore[k].empty()) us.insert(seqStore[k][0]);   // a duplicate: no new element
      std::string cs;
      const int rows = torali::msa(c, us, cs);
      wr(out, (int32_t)us.size());
      wr(out, (int32_t)rows);
      wrs(out, cs);
    }
  }
  out.close();
  return out ? 0 : 2;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "abi") {
    int32_t v[4];
    dellyhip_abi_info(v);
    std::printf("%d %d %d %d\n", v[0], v[1], v[2], v[3]);
    return (v[1] == (int)sizeof(dellyhip_params) && v[2] == (int)sizeof(dellyhip_junction) && v[3] == (int)sizeof(dellyhip_result)) ? 0 : 1;
  }
  if (argc >= 2 && std::string(argv[1]) == "nodevice") {
    try {
      TestConfig c = config_from(dellyhip_params{5, -4, -10, -1, 2, 13, 1000, 100, 0.95f, 0});
      ReadSet rs{"ACGTACGTACGTACGTACGT", "ACGTACGTACGAACGTACGT"};
      std::string cs;
      torali::msa(c, rs, cs);
    } catch (dellyhip_dropin::Error const& e) {
      std::printf("loud failure: %s\n", e.what());
      return e.code == DELLYHIP_E_NODEVICE ? 0 : 3;
    }
    std::printf("msa() returned without a device?\n");
    return 1;
  }
  if (argc == 4 && std::string(argv[1]) == "run") {
    try {
      return run(argv[2], argv[3]);
    } catch (std::exception const& e) {
      std::fprintf(stderr, "dropin_test: %s\n", e.what());
      return 4;
    }
  }
  std::fprintf(stderr, "usage: dropin_test abi | nodevice | run <in.bin> <out.bin>\n");
  return 64;
}
