// y = alpha * A * x on the GPU through the reference's operator API (cf. the call shape of
// /root/reference/examples/device/device_spmv.cpp:36-59): build views over device arrays, inspect once,
// multiply, copy back, compare with a host loop.
#include <cmath>

#include "common.hpp"

int main() {
  using T = float;
  const int m = 200000, n = 150000, per_row = 12;
  auto h = ex::random_csr<T>(m, n, per_row, 1);
  ex::device_csr<T> a(h);
  std::vector<T> x(n);
  for (int j = 0; j < n; ++j)
    x[j] = T(1) + T(j % 7) * T(0.125);
  ex::device_array<T> d_x(x), d_y(static_cast<std::size_t>(m));

  const T alpha = 2.5f;
  auto info = spblas::multiply_inspect(a.view, d_x.span(), d_y.span());       // device-side analysis, once
  spblas::multiply(info, spblas::scaled(alpha, a.view), d_x.span(), d_y.span());  // y = alpha * A * x
  const auto y = d_y.to_host();

  double worst = 0;
  for (int i = 0; i < m; ++i) {
    double ref = 0, mag = 0;
    for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
      ref += double(alpha) * h.values[p] * x[h.colind[p]];
      mag += std::abs(double(alpha) * h.values[p] * x[h.colind[p]]);
    }
    worst = std::max(worst, std::abs(ref - y[i]) / (mag + 1e-30));
  }
  std::printf("device_spmv: %d x %d, nnz %d, max norm-wise error %.3g\n", m, n, (int) h.nnz, worst);
  return worst < 1e-6 ? 0 : 1;
}
