/* edlib.h -- drop-in for the C API of the reference's vendored edlib (src/edlib.h:25-271, src/edlib.cpp:139-300,
 * :1476-1480) on the MI355X: include-path shadowing puts this header in front of src/edlib.h, src/edlib.cpp is not
 * compiled.  The enums, the configuration / result structs and the five functions keep the reference's names, field
 * order and ownership contract (the result's arrays are malloc()ed here and freed by edlibFreeAlignResult or free()),
 * so every call site -- src/split.h:485-568, src/assemble.h:342-693, src/coverage.h:111, src/genotype.h:23,
 * src/merge.h:217, src/svanno.h:154-209, src/util.h:86-150 -- compiles unchanged.
 *
 * edlibAlign runs ONE alignment per call through dellyhip_edlib_align_full (a one-wavefront kernel): it exists so that
 * serial call sites give the reference's result, not to fill a GPU; the batched entry points (dellyhip_refine_batch*,
 * dellyhip_classify_reads, dellyhip_edit_distance_nw_batch) are the product path for the loops around them.
 *
 * Differences from the reference, all reported through result.status = EDLIB_STATUS_ERROR (the reference's own error
 * channel) instead of a wrong answer:
 *   - additionalEqualities: NULL / 0, or exactly the 20 extended-IUPAC pairs of msaEdlib / msaWfa (src/assemble.h:425,
 *     any order); no other set has a caller in the reference;
 *   - target longer than 32 766 or query longer than 32 000 letters;
 *   - no usable gfx950 device / a HIP error (there is no CPU path).
 */
#ifndef DELLYHIP_DROPIN_EDLIB_H
#define DELLYHIP_DROPIN_EDLIB_H

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dellyhip_dropin.h"

#define EDLIB_STATUS_OK 0
#define EDLIB_STATUS_ERROR 1

/* gaps before / after the query: NW global, SHW prefix (free after), HW infix (free before and after) -- src/edlib.h:38-66 */
typedef enum { EDLIB_MODE_NW, EDLIB_MODE_SHW, EDLIB_MODE_HW } EdlibAlignMode;
/* what to compute -- src/edlib.h:71-75 */
typedef enum { EDLIB_TASK_DISTANCE, EDLIB_TASK_LOC, EDLIB_TASK_PATH } EdlibAlignTask;
/* src/edlib.h:82-85 */
typedef enum { EDLIB_CIGAR_STANDARD, EDLIB_CIGAR_EXTENDED } EdlibCigarFormat;

#define EDLIB_EDOP_MATCH 0
#define EDLIB_EDOP_INSERT 1   /* consumes a query letter only */
#define EDLIB_EDOP_DELETE 2   /* consumes a target letter only */
#define EDLIB_EDOP_MISMATCH 3

typedef struct {   /* src/edlib.h:96-99 */
  char first;
  char second;
} EdlibEqualityPair;

typedef struct {   /* src/edlib.h:104-140 */
  int k;           /* >= 0: distances beyond k are reported as -1 */
  EdlibAlignMode mode;
  EdlibAlignTask task;
  const EdlibEqualityPair* additionalEqualities;
  int additionalEqualitiesLength;
} EdlibAlignConfig;

typedef struct {   /* src/edlib.h:160-217 */
  int status;
  int editDistance;
  int* endLocations;
  int* startLocations;
  int numLocations;
  unsigned char* alignment;
  int alignmentLength;
  int alphabetLength;
} EdlibAlignResult;

inline EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task,
                                            const EdlibEqualityPair* additionalEqualities, int additionalEqualitiesLength) {
  EdlibAlignConfig c;
  c.k = k;
  c.mode = mode;
  c.task = task;
  c.additionalEqualities = additionalEqualities;
  c.additionalEqualitiesLength = additionalEqualitiesLength;
  return c;
}

inline EdlibAlignConfig edlibDefaultAlignConfig(void) { return edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0); }

inline void edlibFreeAlignResult(EdlibAlignResult result) {
  if (result.endLocations) free(result.endLocations);
  if (result.startLocations) free(result.startLocations);
  if (result.alignment) free(result.alignment);
}

namespace dellyhip_dropin {
/* 0: no additional equalities, 1: the extended-IUPAC set (src/assemble.h:425), -1: anything else */
inline int equality_kind(const EdlibEqualityPair* eq, int n) {
  if (!eq || n == 0) return 0;
  static const char want[20][2] = {{'M', 'A'}, {'M', 'C'}, {'R', 'A'}, {'R', 'G'}, {'W', 'A'}, {'W', 'T'}, {'B', 'A'}, {'B', '-'}, {'S', 'C'}, {'S', 'G'},
                                   {'Y', 'C'}, {'Y', 'T'}, {'D', 'C'}, {'D', '-'}, {'K', 'G'}, {'K', 'T'}, {'E', 'G'}, {'E', '-'}, {'F', 'T'}, {'F', '-'}};
  if (n != 20) return -1;
  bool seen[20] = {false};
  for (int i = 0; i < n; ++i) {
    int hit = -1;
    for (int j = 0; j < 20 && hit < 0; ++j)
      if (!seen[j] && ((eq[i].first == want[j][0] && eq[i].second == want[j][1]) || (eq[i].first == want[j][1] && eq[i].second == want[j][0]))) hit = j;
    if (hit < 0) return -1;
    seen[hit] = true;
  }
  return 1;
}
}  // namespace dellyhip_dropin

inline EdlibAlignResult edlibAlign(const char* query, int queryLength, const char* target, int targetLength, const EdlibAlignConfig config) {
  namespace dd = dellyhip_dropin;
  EdlibAlignResult r;
  r.status = EDLIB_STATUS_OK;
  r.editDistance = -1;
  r.endLocations = r.startLocations = NULL;
  r.numLocations = 0;
  r.alignment = NULL;
  r.alignmentLength = 0;
  /* distinct letters of both sequences (transformSequences, src/edlib.cpp:152-155) */
  bool used[256] = {false};
  int distinct = 0;
  for (int i = 0; i < queryLength; ++i)
    if (!used[(unsigned char)query[i]]) { used[(unsigned char)query[i]] = true; ++distinct; }
  for (int i = 0; i < targetLength; ++i)
    if (!used[(unsigned char)target[i]]) { used[(unsigned char)target[i]] = true; ++distinct; }
  r.alphabetLength = distinct;
  const int eq = dd::equality_kind(config.additionalEqualities, config.additionalEqualitiesLength);
  const int mode = (int)config.mode, task = (int)config.task;
  if (eq < 0 || mode < 0 || mode > 2 || task < 0 || task > 2 || queryLength < 0 || targetLength < 0) {
    r.status = EDLIB_STATUS_ERROR;
    return r;
  }
  try {
    dellyhip_params p;
    dellyhip_default_params_sr(&p);
    dd::Session& S = dd::session(p);
    const int lcap = targetLength + 1;
    std::vector<int32_t> ends((std::size_t)lcap), starts((std::size_t)lcap);
    std::vector<unsigned char> ops((std::size_t)queryLength + (std::size_t)targetLength + 64);
    int32_t ed = -1, nloc = 0, nops = 0;
    dd::check(dellyhip_edlib_align_full(S.ctx, query, queryLength, target, targetLength, config.k, mode, task, eq, &ed, &nloc, ends.data(),
                                        starts.data(), lcap, ops.data(), (int32_t)ops.size(), &nops));
    r.editDistance = ed;
    if (ed < 0) return r;   /* beyond k: no locations, no alignment */
    r.numLocations = nloc;
    r.endLocations = static_cast<int*>(malloc(sizeof(int) * (std::size_t)(nloc > 0 ? nloc : 1)));
    for (int i = 0; i < nloc; ++i) r.endLocations[i] = ends[(std::size_t)i];
    const bool empty = queryLength == 0 || targetLength == 0;   /* src/edlib.cpp:160-178: end location only */
    if (task >= 1 && !empty) {
      r.startLocations = static_cast<int*>(malloc(sizeof(int) * (std::size_t)(nloc > 0 ? nloc : 1)));
      for (int i = 0; i < nloc; ++i) r.startLocations[i] = starts[(std::size_t)i];
    }
    if (task == 2 && !empty) {
      r.alignment = static_cast<unsigned char*>(malloc((std::size_t)(nops > 0 ? nops : 1)));
      std::memcpy(r.alignment, ops.data(), (std::size_t)nops);
      r.alignmentLength = nops;
    }
  } catch (dd::Error const&) {
    edlibFreeAlignResult(r);
    r.endLocations = r.startLocations = NULL;
    r.alignment = NULL;
    r.numLocations = r.alignmentLength = 0;
    r.editDistance = -1;
    r.status = EDLIB_STATUS_ERROR;
  }
  return r;
}

/* run-length text of an alignment; '=' / 'X' in the extended format, 'M' for both in the standard one; malloc()ed,
 * NULL for an unknown format or an op code beyond 3 (src/edlib.h:248-271) */
inline char* edlibAlignmentToCigar(const unsigned char* alignment, int alignmentLength, EdlibCigarFormat cigarFormat) {
  if (cigarFormat != EDLIB_CIGAR_EXTENDED && cigarFormat != EDLIB_CIGAR_STANDARD) return NULL;
  const char ext[4] = {'=', 'I', 'D', 'X'}, std_[4] = {'M', 'I', 'D', 'M'};
  const char* letter = (cigarFormat == EDLIB_CIGAR_EXTENDED) ? ext : std_;
  std::string out;
  int i = 0;
  while (i < alignmentLength) {
    if (alignment[i] > 3) return NULL;
    const char ch = letter[alignment[i]];
    int j = i;
    while (j < alignmentLength && alignment[j] <= 3 && letter[alignment[j]] == ch) ++j;
    out += std::to_string(j - i);
    out += ch;
    i = j;
  }
  char* c = static_cast<char*>(malloc(out.size() + 1));
  std::memcpy(c, out.c_str(), out.size() + 1);
  return c;
}

#endif
