// Shared by the example programs: a tiny RAII device buffer and a random CSR matrix on the host.
// (The reference's examples use thrust::device_vector and spblas::generate_csr; neither is needed here.)
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <span>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <spblas_gfx950/spblas.hpp>

namespace ex {

inline void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    std::exit(2);
  }
}

template <typename T>
class device_array {
public:
  explicit device_array(std::size_t n) : n_(n) {
    hip_ok(hipMalloc(reinterpret_cast<void**>(&p_), std::max<std::size_t>(n, 1) * sizeof(T)), "hipMalloc");
  }
  explicit device_array(const std::vector<T>& host) : device_array(host.size()) {
    if (!host.empty())
      hip_ok(hipMemcpy(p_, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  device_array(const device_array&) = delete;
  ~device_array() {
    (void) hipFree(p_);
  }
  T* data() const {
    return p_;
  }
  std::size_t size() const {
    return n_;
  }
  std::span<T> span() const {
    return std::span<T>(p_, n_);
  }
  std::vector<T> to_host() const {
    std::vector<T> h(n_);
    if (n_)
      hip_ok(hipMemcpy(h.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return h;
  }

private:
  T* p_ = nullptr;
  std::size_t n_ = 0;
};

template <typename T>
struct host_csr {
  std::vector<T> values;
  std::vector<spblas::offset_t> rowptr;
  std::vector<spblas::index_t> colind;
  spblas::index<spblas::index_t> shape;
  spblas::offset_t nnz;
};

// m x n matrix with `per_row` entries in every row, columns uniform (unsorted, repeats possible)
template <typename T>
host_csr<T> random_csr(int m, int n, int per_row, unsigned seed) {
  std::mt19937 g(seed);
  std::uniform_int_distribution<int> col(0, n - 1);
  std::uniform_real_distribution<T> val(T(0.5), T(1.5));
  host_csr<T> a;
  a.shape = spblas::index<spblas::index_t>(m, n);
  a.rowptr.resize(m + 1);
  for (int i = 0; i <= m; ++i)
    a.rowptr[i] = i * per_row;
  a.nnz = m * per_row;
  a.colind.resize(a.nnz);
  a.values.resize(a.nnz);
  for (auto& c : a.colind)
    c = col(g);
  for (auto& v : a.values)
    v = val(g);
  return a;
}

template <typename T>
struct device_csr {
  device_array<T> values;
  device_array<spblas::offset_t> rowptr;
  device_array<spblas::index_t> colind;
  spblas::csr_view<T, spblas::index_t, spblas::offset_t> view;
  explicit device_csr(const host_csr<T>& h)
      : values(h.values), rowptr(h.rowptr), colind(h.colind),
        view(values.data(), rowptr.data(), colind.data(), h.shape, h.nnz) {}
};

} // namespace ex
