// TEST ONLY -- compiled C++ caller of include/delly_dropin/edlib.h, the drop-in for the reference's vendored edlib C API
// (src/edlib.h:146-271).  It includes "edlib.h" exactly as the reference's call sites do (the include path puts the drop-in in
// front of $(REF)/src; src/edlib.cpp is NOT compiled in), calls edlibAlign / edlibAlignmentToCigar / edlibFreeAlignResult with
// the reference's call shapes on a case file written by tests/test_gpu_edlib_dropin.py, and writes every field of each
// EdlibAlignResult to a result file that the Python side compares with oracle/_ref (the reference's real edlib).
//
//   edlib_dropin_test run <in.bin> <out.bin>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "edlib.h"   // include/delly_dropin/edlib.h (NOT the reference's)

namespace {
template <typename T>
bool rd(std::ifstream& f, T& v) { return (bool)f.read(reinterpret_cast<char*>(&v), sizeof v); }
template <typename T>
void wr(std::ofstream& f, T const& v) { f.write(reinterpret_cast<const char*>(&v), sizeof v); }

// the two helpers of src/util.h:86-99 the path uses, with the reference's statements (they only touch the C API)
int32_t infixStartLike(EdlibAlignResult& cigar) { return cigar.startLocations ? cigar.startLocations[0] : -1; }
}  // namespace

int main(int argc, char** argv) {
  if (argc != 4 || std::string(argv[1]) != "run") {
    std::fprintf(stderr, "usage: edlib_dropin_test run <in.bin> <out.bin>\n");
    return 2;
  }
  std::ifstream in(argv[2], std::ios::binary);
  std::ofstream out(argv[3], std::ios::binary);
  int32_t n = 0;
  if (!rd(in, n)) return 2;
  EdlibEqualityPair iupac[20] = {{'M', 'A'}, {'M', 'C'}, {'R', 'A'}, {'R', 'G'}, {'W', 'A'}, {'W', 'T'}, {'B', 'A'}, {'B', '-'}, {'S', 'C'}, {'S', 'G'},
                                 {'Y', 'C'}, {'Y', 'T'}, {'D', 'C'}, {'D', '-'}, {'K', 'G'}, {'K', 'T'}, {'E', 'G'}, {'E', '-'}, {'F', 'T'}, {'F', '-'}};
  EdlibEqualityPair other[1] = {{'a', 'A'}};   // a set the drop-in does not serve: must come back as EDLIB_STATUS_ERROR
  for (int32_t i = 0; i < n; ++i) {
    int32_t k, mode, task, eq, qn, tn;
    if (!rd(in, k) || !rd(in, mode) || !rd(in, task) || !rd(in, eq) || !rd(in, qn) || !rd(in, tn)) return 2;
    std::string q((size_t)qn, '\0'), t((size_t)tn, '\0');
    in.read(&q[0], qn);
    in.read(&t[0], tn);
    const EdlibEqualityPair* pairs = (eq == 1) ? iupac : ((eq == 2) ? other : NULL);
    EdlibAlignResult r = edlibAlign(q.c_str(), (int)q.size(), t.c_str(), (int)t.size(),
                                    edlibNewAlignConfig(k, (EdlibAlignMode)mode, (EdlibAlignTask)task, pairs, eq == 1 ? 20 : (eq == 2 ? 1 : 0)));
    wr(out, (int32_t)r.status);
    wr(out, (int32_t)r.editDistance);
    wr(out, (int32_t)r.numLocations);
    wr(out, (int32_t)r.alignmentLength);
    wr(out, (int32_t)r.alphabetLength);
    wr(out, (int32_t)(r.endLocations ? 1 : 0));
    wr(out, (int32_t)(r.startLocations ? 1 : 0));
    wr(out, (int32_t)(r.alignment ? 1 : 0));
    wr(out, (int32_t)infixStartLike(r));
    if (r.status == EDLIB_STATUS_OK) {
      for (int j = 0; j < r.numLocations; ++j) wr(out, (int32_t)(r.endLocations ? r.endLocations[j] : -9));
      for (int j = 0; j < r.numLocations; ++j) wr(out, (int32_t)(r.startLocations ? r.startLocations[j] : -9));
      if (r.alignment) out.write(reinterpret_cast<const char*>(r.alignment), r.alignmentLength);
      for (int fmt = 0; fmt < 2; ++fmt) {
        char* c = r.alignment ? edlibAlignmentToCigar(r.alignment, r.alignmentLength, (EdlibCigarFormat)fmt) : NULL;
        const int32_t len = c ? (int32_t)std::strlen(c) : -1;
        wr(out, len);
        if (c) { out.write(c, len); free(c); }
      }
    }
    edlibFreeAlignResult(r);
  }
  EdlibAlignConfig d = edlibDefaultAlignConfig();
  wr(out, (int32_t)d.k);
  wr(out, (int32_t)d.mode);
  wr(out, (int32_t)d.task);
  return 0;
}
