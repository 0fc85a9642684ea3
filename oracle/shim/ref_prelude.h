// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not part of the product.
//
// Declarations the reference's hot-path headers expect from htslib / Boost /
// src/util.h, so that tags.h, edlib.h, msa.h (align.h gotoh.h needle.h) and
// split.h compile UNMODIFIED from /root/reference/src.
// Only containers, PODs and four tiny util.h helpers are supplied here; every
// arithmetic statement on the path is the reference's own.
#ifndef DELLY_ORACLE_REF_PRELUDE_H
#define DELLY_ORACLE_REF_PRELUDE_H

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include <boost/dynamic_bitset.hpp>
#include <boost/multi_array.hpp>

// ---- htslib PODs / flags referenced by src/tags.h and src/msa.h ----------
#define BAM_FREVERSE 16
#define BAM_FMREVERSE 32

typedef struct {
  int32_t n_targets;
  uint32_t* target_len;
  char** target_name;
} bam_hdr_t;

typedef struct {
  int64_t pos;
  int32_t tid;
  uint16_t flag;
  int32_t mtid;
  int64_t mpos;
  int64_t isize;
} bam1_core_t;

typedef struct {
  bam1_core_t core;
  uint8_t* data;
} bam1_t;

inline char* bam_get_qname(bam1_t* b) { return reinterpret_cast<char*>(b->data); }

// ---- Boost string/tokenizer bits (src/split.h uses to_upper_copy; the
// tokenizer/lexical_cast names only occur inside the never-instantiated
// _alignmentScore template of src/align.h:231-245) -------------------------
namespace boost {

inline std::string to_upper_copy(std::string const& s) {
  std::string r(s);
  for (std::size_t i = 0; i < r.size(); ++i) r[i] = (char)std::toupper((unsigned char)r[i]);
  return r;
}

template <typename TChar>
struct char_separator {
  explicit char_separator(const TChar*) {}
};

template <typename TSep>
struct tokenizer {
  typedef std::vector<std::string>::iterator iterator;
  template <typename TStr>
  tokenizer(TStr const&, TSep const&) {}
  iterator begin() { return v_.begin(); }
  iterator end() { return v_.end(); }
  std::vector<std::string> v_;
};

template <typename T, typename S>
inline T lexical_cast(S const&) { return T(); }

}  // namespace boost

#endif
