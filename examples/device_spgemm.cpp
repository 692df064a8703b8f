// C = A * B with the two-phase device API (cf. /root/reference/examples/simple_spgemm.cpp:52-60 and
// test/gtest/device/spgemm_test.cpp:37-53): multiply_compute sizes C, the caller allocates, multiply_fill
// writes columns (ascending) and values; then D = C + A*B via add() as a second operation.
#include <map>

#include "common.hpp"

int main() {
  using T = float;
  using I = spblas::index_t;
  using O = spblas::offset_t;
  const int m = 3000, k = 2500, n = 2000;
  auto ha = ex::random_csr<T>(m, k, 6, 2), hb = ex::random_csr<T>(k, n, 5, 3);
  ex::device_csr<T> a(ha), b(hb);

  ex::device_array<O> c_rowptr(static_cast<std::size_t>(m + 1));
  spblas::csr_view<T, I, O> c(nullptr, c_rowptr.data(), nullptr, {m, n}, 0);
  auto info = spblas::multiply_compute(a.view, b.view, c);      // symbolic: rowptr + nnz
  const auto nnz = info.result_nnz();
  ex::device_array<T> c_values(static_cast<std::size_t>(nnz));
  ex::device_array<I> c_colind(static_cast<std::size_t>(nnz));
  c.update(c_values.span(), c_rowptr.span(), c_colind.span(), {m, n}, (O) nnz);
  spblas::multiply_fill(info, a.view, b.view, c);                // numeric

  // check against a host SPA, row by row
  const auto rp = c_rowptr.to_host();
  const auto ci = c_colind.to_host();
  const auto cv = c_values.to_host();
  long bad = 0;
  for (int i = 0; i < m; ++i) {
    std::map<I, double> ref;
    for (auto p = ha.rowptr[i]; p < ha.rowptr[i + 1]; ++p)
      for (auto q = hb.rowptr[ha.colind[p]]; q < hb.rowptr[ha.colind[p] + 1]; ++q)
        ref[hb.colind[q]] += double(ha.values[p]) * hb.values[q];
    auto it = ref.begin();
    bad += (long) ref.size() != rp[i + 1] - rp[i];
    for (auto p = rp[i]; p < rp[i + 1] && it != ref.end(); ++p, ++it)
      bad += ci[p] != it->first || std::abs(cv[p] - it->second) > 1e-5 * std::abs(it->second);
  }

  // E = 2*C + C  (add of two CSR matrices with scaled views)
  ex::device_array<O> e_rowptr(static_cast<std::size_t>(m + 1));
  spblas::csr_view<T, I, O> e(nullptr, e_rowptr.data(), nullptr, {m, n}, 0);
  auto add_info = spblas::add_inspect(spblas::scaled(2.0f, c), c, e);
  ex::device_array<T> e_values(static_cast<std::size_t>(add_info.result_nnz()));
  ex::device_array<I> e_colind(static_cast<std::size_t>(add_info.result_nnz()));
  e.update(e_values.span(), e_rowptr.span(), e_colind.span(), {m, n}, (O) add_info.result_nnz());
  spblas::add_compute(add_info, spblas::scaled(2.0f, c), c, e);
  const auto ev = e_values.to_host();
  bad += add_info.result_nnz() != nnz;
  for (std::size_t p = 0; p < ev.size() && p < cv.size(); ++p)
    bad += std::abs(ev[p] - 3.0f * cv[p]) > 1e-5f * std::abs(cv[p]);

  std::printf("device_spgemm: C %d x %d nnz %lld, mismatches %ld\n", m, n, (long long) nnz, bad);
  return bad == 0 ? 0 : 1;
}
