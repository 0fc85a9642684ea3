// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not part of the product.
//
// Minimal, independently written stand-in for the subset of boost::multi_array
// that the reference's hot-path headers use (src/align.h, src/gotoh.h,
// src/needle.h, src/msa.h, src/split.h, src/assemble.h:1-733):
//   multi_array<T,2> a;  multi_array<T,2> a(boost::extents[r][c]);
//   a.resize(boost::extents[r][c]);  a.shape()[i];  a[i][j];  copy/assign;
//   typedef multi_array<T,2>::index.
// Semantics kept from Boost: value-initialised storage, resize preserves the
// overlapping block and value-initialises the rest (longestHomology and the
// rev-glue in needle.h:19,209-217 rely on zero-init).
// Boost itself is not installed in this image.
#ifndef DELLY_ORACLE_SHIM_MULTI_ARRAY_HPP
#define DELLY_ORACLE_SHIM_MULTI_ARRAY_HPP

#include <cstddef>
#include <vector>
#include <algorithm>

namespace boost {

namespace shim_detail {
struct extent2 {
  std::size_t d0, d1;
};
struct extent1 {
  std::size_t d0;
  extent2 operator[](std::size_t d1) const { return extent2{d0, d1}; }
};
struct extent_gen {
  extent1 operator[](std::size_t d0) const { return extent1{d0}; }
};
}  // namespace shim_detail

static const shim_detail::extent_gen extents = shim_detail::extent_gen();

template <typename T, std::size_t NDims>
class multi_array;

template <typename T>
class multi_array<T, 2> {
 public:
  typedef std::ptrdiff_t index;
  typedef std::size_t size_type;
  typedef T element;

  // vector<bool> is bit-packed and has no T* rows: store bools as bytes.
  struct bool_byte {
    bool v;
    bool_byte() : v(false) {}
  };

  multi_array() { shape_[0] = 0; shape_[1] = 0; }
  explicit multi_array(shim_detail::extent2 const& e) : data_(e.d0 * e.d1) {
    shape_[0] = e.d0;
    shape_[1] = e.d1;
  }

  void resize(shim_detail::extent2 const& e) {
    std::vector<T> nd(e.d0 * e.d1);
    std::size_t r = std::min(e.d0, shape_[0]);
    std::size_t c = std::min(e.d1, shape_[1]);
    for (std::size_t i = 0; i < r; ++i)
      for (std::size_t j = 0; j < c; ++j) nd[i * e.d1 + j] = data_[i * shape_[1] + j];
    data_.swap(nd);
    shape_[0] = e.d0;
    shape_[1] = e.d1;
  }

  const size_type* shape() const { return shape_; }

  T* operator[](index i) { return data_.data() + static_cast<std::size_t>(i) * shape_[1]; }
  const T* operator[](index i) const { return data_.data() + static_cast<std::size_t>(i) * shape_[1]; }

 private:
  std::vector<T> data_;
  size_type shape_[2];
};

// bool specialisation: plain byte storage so that operator[] yields bool*.
template <>
class multi_array<bool, 2> {
 public:
  typedef std::ptrdiff_t index;
  typedef std::size_t size_type;
  typedef bool element;

  multi_array() : data_(0), n_(0) { shape_[0] = 0; shape_[1] = 0; }
  explicit multi_array(shim_detail::extent2 const& e) : data_(0), n_(0) {
    shape_[0] = 0; shape_[1] = 0;
    resize(e);
  }
  multi_array(multi_array const& o) : data_(0), n_(0) {
    shape_[0] = 0; shape_[1] = 0;
    *this = o;
  }
  multi_array& operator=(multi_array const& o) {
    if (this != &o) {
      delete[] data_;
      n_ = o.n_;
      data_ = n_ ? new bool[n_] : 0;
      for (std::size_t i = 0; i < n_; ++i) data_[i] = o.data_[i];
      shape_[0] = o.shape_[0];
      shape_[1] = o.shape_[1];
    }
    return *this;
  }
  ~multi_array() { delete[] data_; }

  void resize(shim_detail::extent2 const& e) {
    std::size_t nn = e.d0 * e.d1;
    bool* nd = nn ? new bool[nn]() : 0;
    std::size_t r = std::min(e.d0, shape_[0]);
    std::size_t c = std::min(e.d1, shape_[1]);
    for (std::size_t i = 0; i < r; ++i)
      for (std::size_t j = 0; j < c; ++j) nd[i * e.d1 + j] = data_[i * shape_[1] + j];
    delete[] data_;
    data_ = nd;
    n_ = nn;
    shape_[0] = e.d0;
    shape_[1] = e.d1;
  }
  const size_type* shape() const { return shape_; }
  bool* operator[](index i) { return data_ + static_cast<std::size_t>(i) * shape_[1]; }
  const bool* operator[](index i) const { return data_ + static_cast<std::size_t>(i) * shape_[1]; }

 private:
  bool* data_;
  std::size_t n_;
  size_type shape_[2];
};

}  // namespace boost

#endif
