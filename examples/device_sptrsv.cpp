// x = inv(L) b with triangular_solve_inspect / triangular_solve (cf. /root/reference/examples/sptrsv_csr.cpp):
// a sparse lower-triangular system with a stored diagonal, solved on the GPU by level sets.
#include <cmath>

#include "common.hpp"

int main() {
  using T = double;
  using I = spblas::index_t;
  using O = spblas::offset_t;
  const int n = 100000, below = 5;
  std::mt19937 g(4);
  ex::host_csr<T> h;
  h.shape = spblas::index<I>(n, n);
  h.rowptr.push_back(0);
  for (int i = 0; i < n; ++i) {
    for (int t = 0; t < below && i > 0; ++t) {
      h.colind.push_back((I) (g() % i));
      h.values.push_back(T(0.1) * T((g() % 100) + 1) / T(100));
    }
    h.colind.push_back(i);
    h.values.push_back(T(2) + T(i % 3));
    h.rowptr.push_back((O) h.colind.size());
  }
  h.nnz = (O) h.colind.size();
  ex::device_csr<T> a(h);
  std::vector<T> b(n);
  for (int i = 0; i < n; ++i)
    b[i] = T(1) + T(i % 11);
  ex::device_array<T> d_b(b), d_x(static_cast<std::size_t>(n));

  auto info = spblas::triangular_solve_inspect(a.view, spblas::lower_triangle, spblas::explicit_diagonal, d_b.span(),
                                               d_x.span());
  spblas::triangular_solve(info, a.view, spblas::lower_triangle, spblas::explicit_diagonal, d_b.span(), d_x.span());
  const auto x = d_x.to_host();

  double worst = 0;  // residual of every row
  for (int i = 0; i < n; ++i) {
    double s = 0, mag = std::abs(b[i]);
    for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
      s += h.values[p] * x[h.colind[p]];
      mag += std::abs(h.values[p] * x[h.colind[p]]);
    }
    worst = std::max(worst, std::abs(s - b[i]) / mag);
  }
  std::printf("device_sptrsv: n %d, nnz %d, max row residual %.3g\n", n, (int) h.nnz, worst);
  return worst < 1e-12 ? 0 : 1;
}
