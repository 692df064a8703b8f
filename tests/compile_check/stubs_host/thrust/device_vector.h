// TEST INFRASTRUCTURE ONLY (tests/compile_check/build_dropin.py: build_reference_device_tests_on_oracle()).
// The reference's DEVICE tests keep their operands in thrust::device_vector.  When those tests are linked to the CPU
// oracle (oracle_shim.c) there is no device: this header stands in for <thrust/device_vector.h> with a host vector that
// offers what the tests use -- construction from a std::vector / a size / another device_vector, data().get(),
// begin() / end(), and thrust::copy.  It is only on the include path of the oracle build (the GPU build of the same
// tests uses the real rocThrust).
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace thrust {

template <typename T>
class device_ptr {
public:
  explicit device_ptr(T* p = nullptr) : p_(p) {}
  T* get() const { return p_; }

private:
  T* p_;
};

template <typename T>
class device_vector {
public:
  device_vector() = default;
  explicit device_vector(std::size_t n) : v_(n) {}
  device_vector(std::size_t n, const T& x) : v_(n, x) {}
  template <typename U>
  device_vector(const std::vector<U>& h) : v_(h.begin(), h.end()) {}
  template <typename U>
  device_vector(const device_vector<U>& o) : v_(o.begin(), o.end()) {}
  device_vector(const device_vector&) = default;
  device_vector& operator=(const device_vector&) = default;
  device_ptr<T> data() { return device_ptr<T>(v_.data()); }
  device_ptr<const T> data() const { return device_ptr<const T>(v_.data()); }
  auto begin() { return v_.begin(); }
  auto end() { return v_.end(); }
  auto begin() const { return v_.begin(); }
  auto end() const { return v_.end(); }
  std::size_t size() const { return v_.size(); }
  T& operator[](std::size_t i) { return v_[i]; }
  const T& operator[](std::size_t i) const { return v_[i]; }

private:
  std::vector<T> v_;
};

template <typename InIt, typename OutIt>
OutIt copy(InIt first, InIt last, OutIt out) {
  return std::copy(first, last, out);
}

} // namespace thrust
