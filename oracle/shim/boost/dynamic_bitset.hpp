// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not part of the product.
//
// Minimal stand-in for boost::dynamic_bitset<> as used by the reference's
// src/gotoh.h:87-91 and src/needle.h:239-241: (n, false) constructor and an
// operator[] returning an assignable / testable bit reference.
#ifndef DELLY_ORACLE_SHIM_DYNAMIC_BITSET_HPP
#define DELLY_ORACLE_SHIM_DYNAMIC_BITSET_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

namespace boost {

template <typename Block = unsigned long>
class dynamic_bitset {
 public:
  class reference {
   public:
    reference(std::uint64_t& w, unsigned b) : w_(w), b_(b) {}
    reference& operator=(bool v) {
      if (v) w_ |= (std::uint64_t(1) << b_);
      else w_ &= ~(std::uint64_t(1) << b_);
      return *this;
    }
    reference& operator=(int v) { return (*this = (v != 0)); }
    operator bool() const { return (w_ >> b_) & 1u; }
   private:
    std::uint64_t& w_;
    unsigned b_;
  };

  dynamic_bitset() : n_(0) {}
  explicit dynamic_bitset(std::size_t n, bool v = false)
      : words_((n + 63) / 64, v ? ~std::uint64_t(0) : 0), n_(n) {}

  std::size_t size() const { return n_; }
  reference operator[](std::size_t i) { return reference(words_[i >> 6], unsigned(i & 63)); }
  bool operator[](std::size_t i) const { return (words_[i >> 6] >> (i & 63)) & 1u; }

 private:
  std::vector<std::uint64_t> words_;
  std::size_t n_;
};

}  // namespace boost

#endif
