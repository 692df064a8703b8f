// The drop-in backend RUNNING behind the reference's own API.
//
// Same translation environment as dropin_check.cpp -- the reference tree's <spblas/spblas.hpp> with
// -DSPBLAS_ENABLE_GFX950 after INTEGRATION.md section 2's edits, include/spblas/vendor/gfx950/*.hpp as the vendor
// backend -- but compiled to a program and linked to libspblas_gfx950.so.  It is built where the reference tree
// exists (spblas-reference_amd/_build.py: build_dropin_run(), called from __graft_entry__.build()); the binary
// (tests/compile_check/_build/dropin_run, git-ignored) travels to the GPU box with the snapshot and
// tests/test_gpu_dropin.py runs it there.  Every operation is called exactly as a user of the reference would
// (spblas::multiply(a, x, y), multiply_compute / multiply_fill, add, transpose, scale, triangular_solve on
// spblas::csr_view over device pointers) and checked against plain host loops written here.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <span>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <spblas/spblas.hpp>

#include <spblas/algorithms/transposed.hpp>

using T = float;
using I = spblas::index_t;
using O = spblas::offset_t;

static int g_checks = 0, g_failed = 0;

#define HIP_OK(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);           \
      std::exit(2);                                                                                         \
    }                                                                                                       \
  } while (0)

template <typename U>
struct dev_array {
  U* p = nullptr;
  std::size_t n = 0;
  dev_array() = default;
  explicit dev_array(std::size_t count) : n(count) { HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), std::max<std::size_t>(count, 1) * sizeof(U))); }
  explicit dev_array(const std::vector<U>& h) : dev_array(h.size()) {
    if (!h.empty())
      HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
  }
  dev_array(const dev_array&) = delete;
  dev_array& operator=(const dev_array&) = delete;
  ~dev_array() { (void) hipFree(p); }
  std::vector<U> host() const {
    std::vector<U> h(n);
    HIP_OK(hipDeviceSynchronize());
    if (n)
      HIP_OK(hipMemcpy(h.data(), p, n * sizeof(U), hipMemcpyDeviceToHost));
    return h;
  }
  std::span<U> span() const { return std::span<U>(p, n); }
};

struct host_csr {
  I m = 0, n = 0;
  std::vector<O> rowptr;
  std::vector<I> colind;
  std::vector<T> values;
  O nnz() const { return static_cast<O>(colind.size()); }
};

static std::uint64_t g_seed = 0x9E3779B97F4A7C15ull;
static std::uint32_t next_u32() {
  g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull;
  return static_cast<std::uint32_t>(g_seed >> 33);
}
static T next_val() { return static_cast<T>(static_cast<int>(next_u32() % 17) - 8) / 4.0f; }  // multiples of 0.25

// `per` DISTINCT sorted columns in every row (so that set-like outputs are unambiguous)
static host_csr random_csr(I m, I n, int per) {
  host_csr a;
  a.m = m;
  a.n = n;
  a.rowptr.assign(static_cast<std::size_t>(m) + 1, 0);
  for (I r = 0; r < m; ++r) {
    std::vector<I> cols;
    const int k = std::min<int>(per, n);
    while (static_cast<int>(cols.size()) < k) {
      const I c = static_cast<I>(next_u32() % static_cast<std::uint32_t>(n));
      if (std::find(cols.begin(), cols.end(), c) == cols.end())
        cols.push_back(c);
    }
    std::sort(cols.begin(), cols.end());
    for (I c : cols) {
      a.colind.push_back(c);
      T v = next_val();
      a.values.push_back(v == 0 ? T(0.5) : v);
    }
    a.rowptr[static_cast<std::size_t>(r) + 1] = static_cast<O>(a.colind.size());
  }
  return a;
}

struct dev_csr {
  dev_array<T> values;
  dev_array<O> rowptr;
  dev_array<I> colind;
  I m, n;
  O nnz;
  explicit dev_csr(const host_csr& h) : values(h.values), rowptr(h.rowptr), colind(h.colind), m(h.m), n(h.n), nnz(h.nnz()) {}
  spblas::csr_view<T, I, O> view() const { return spblas::csr_view<T, I, O>(values.p, rowptr.p, colind.p, {m, n}, nnz); }
};

static void expect(bool ok, const char* what) {
  ++g_checks;
  if (!ok) {
    ++g_failed;
    std::fprintf(stderr, "FAILED: %s\n", what);
  }
}

static bool close_vec(const std::vector<T>& got, const std::vector<double>& want, const std::vector<double>& scale) {
  if (got.size() != want.size())
    return false;
  for (std::size_t i = 0; i < got.size(); ++i) {
    const double tol = 1e-6 * (scale.empty() ? std::fabs(want[i]) + 1.0 : scale[i] + 1e-30);
    if (!(std::fabs(static_cast<double>(got[i]) - want[i]) <= tol))
      return false;
  }
  return true;
}

static void host_spmv(const host_csr& a, const std::vector<T>& x, double alpha, std::vector<double>& y, std::vector<double>& absrow) {
  y.assign(a.m, 0.0);
  absrow.assign(a.m, 0.0);
  for (I r = 0; r < a.m; ++r)
    for (O p = a.rowptr[r]; p < a.rowptr[r + 1]; ++p) {
      y[r] += alpha * a.values[p] * x[a.colind[p]];
      absrow[r] += std::fabs(alpha * a.values[p] * x[a.colind[p]]);
    }
}

static host_csr host_transpose(const host_csr& a) {
  host_csr t;
  t.m = a.n;
  t.n = a.m;
  t.rowptr.assign(static_cast<std::size_t>(a.n) + 1, 0);
  for (I c : a.colind)
    ++t.rowptr[static_cast<std::size_t>(c) + 1];
  for (I j = 0; j < a.n; ++j)
    t.rowptr[j + 1] += t.rowptr[j];
  t.colind.resize(a.colind.size());
  t.values.resize(a.values.size());
  std::vector<O> cur(t.rowptr.begin(), t.rowptr.end() - 1);
  for (I r = 0; r < a.m; ++r)
    for (O p = a.rowptr[r]; p < a.rowptr[r + 1]; ++p) {
      const O q = cur[a.colind[p]]++;
      t.colind[q] = r;
      t.values[q] = a.values[p];
    }
  return t;
}

int main() {
  using namespace spblas;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "dropin_run: no HIP device\n");
    return 3;
  }

  // ---- SpMV: multiply(a, x, y), inspected, matrix_opt, scaled, csc_view, transposed(a) ----
  {
    const I m = 20000, n = 30000;
    const host_csr ha = random_csr(m, n, 12);
    dev_csr da(ha);
    std::vector<T> hx(n);
    for (auto& v : hx)
      v = next_val();
    dev_array<T> dx(hx), dy(static_cast<std::size_t>(m));
    auto a = da.view();
    std::span<T> x = dx.span(), y = dy.span();
    std::vector<double> want, absrow;
    host_spmv(ha, hx, 1.0, want, absrow);
    multiply(a, x, y);
    expect(close_vec(dy.host(), want, absrow), "multiply(a, x, y)");
    HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
    operation_info_t info = multiply_inspect(a, x, y);
    multiply(info, a, x, y);
    expect(close_vec(dy.host(), want, absrow), "multiply(info, a, x, y) after multiply_inspect");
    HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
    matrix_opt a_opt(a);
    operation_info_t info_opt = multiply_inspect(a_opt, x, y);
    multiply(info_opt, a_opt, x, y);
    expect(close_vec(dy.host(), want, absrow), "multiply(info, matrix_opt(a), x, y)");
    {
      // the plan lives in the matrix_opt (views/matrix_opt_impl.hpp:25-28,90-92 is where oneMKL keeps its handle): a
      // multiply WITHOUT info, also through a view stacked on a copy of a_opt, reuses it -- one plan built, ever
      auto* cached = a_opt.gfx950_state_->get<__gfx950::spmv_state_t>();
      expect(cached != nullptr && cached->plans_built() == 1 && cached->plan() != nullptr,
             "multiply_inspect(matrix_opt(a), ...) keeps the plan in the view");
      HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
      multiply(a_opt, x, y);
      expect(close_vec(dy.host(), want, absrow), "multiply(matrix_opt(a), x, y) without info");
      std::vector<double> want2, abs2;
      host_spmv(ha, hx, 3.0, want2, abs2);
      multiply(scaled(3.0f, a_opt), x, y);
      expect(close_vec(dy.host(), want2, abs2), "multiply(scaled(alpha, matrix_opt(a)), x, y) without info");
      std::int64_t pinfo[12] = {0};
      expect(spblas_gfx950_plan_info(cached->plan(), pinfo) == SPBLAS_GFX950_STATUS_SUCCESS && pinfo[0] != 0 &&
                 a_opt.gfx950_state_->get<__gfx950::spmv_state_t>() == cached && cached->plans_built() == 1,
             "second and third call did not plan again (spblas_gfx950_plan_info on the cached plan)");
    }
    host_spmv(ha, hx, -2.5, want, absrow);
    multiply(scaled(-2.5f, a), x, y);
    expect(close_vec(dy.host(), want, absrow), "multiply(scaled(alpha, a), x, y)");
    multiply(a, scaled(-2.5f, x), y);
    expect(close_vec(dy.host(), want, absrow), "multiply(a, scaled(alpha, x), y)");
    // the same matrix as csc_view: colptr / rowind of A are rowptr / colind of A^T
    const host_csr ht = host_transpose(ha);
    dev_csr dt(ht);
    csc_view<T, I, O> a_csc(dt.values.p, dt.rowptr.p, dt.colind.p, {m, n}, ha.nnz());
    host_spmv(ha, hx, 1.0, want, absrow);
    multiply(a_csc, x, y);
    expect(close_vec(dy.host(), want, absrow), "multiply(csc_view, x, y)");
    {
      // inspected PLAIN csc_view: structure only -- every multiply reads the caller's arrays (op = T), so values
      // rewritten in place after the inspect are seen (algorithms/multiply_impl.hpp:48-52 reads A on every call)
      const std::vector<T> y_atomics = dy.host();
      HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
      operation_info_t info_csc = multiply_inspect(a_csc, x, y);
      expect(info_csc.state_.get_state<__gfx950::csc_spmv_state_t<T>>() == nullptr,
             "multiply_inspect(csc_view, ...) keeps no copy of the values");
      multiply(info_csc, a_csc, x, y);
      expect(close_vec(dy.host(), want, absrow), "multiply(info, csc_view, x, y) after multiply_inspect");
      {
        std::vector<T> doubled(ht.values);
        for (auto& v : doubled)
          v *= 2;
        HIP_OK(hipMemcpy(dt.values.p, doubled.data(), doubled.size() * sizeof(T), hipMemcpyHostToDevice));
        std::vector<double> want2, abs2;
        host_spmv(ha, hx, 2.0, want2, abs2);
        multiply(info_csc, a_csc, x, y);
        expect(close_vec(dy.host(), want2, abs2), "inspected csc_view: values rewritten IN PLACE are seen by the next multiply");
        HIP_OK(hipMemcpy(dt.values.p, ht.values.data(), ht.values.size() * sizeof(T), hipMemcpyHostToDevice));
      }
      // matrix_opt over the csc_view: the row-major form is materialised once (csc_spmv_state_t) and the regular kernels
      // run; the result equals the atomics path above to rounding (vendor/rocsparse/detail/get_transpose.hpp:19-29)
      matrix_opt c_opt(a_csc);
      multiply_inspect(c_opt, x, y);
      auto* st = c_opt.gfx950_state_->get<__gfx950::csc_spmv_state_t<T>>();
      expect(st != nullptr && st->inspections() == 1, "multiply_inspect(matrix_opt(csc_view), ...) materialises the transpose once");
      HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
      multiply(c_opt, x, y);
      multiply(c_opt, x, y);
      const std::vector<T> y_insp = dy.host();
      expect(close_vec(y_insp, want, absrow) && st != nullptr && st->inspections() == 1,
             "multiply(matrix_opt(csc_view), x, y) without info");
      bool same = true;
      for (std::size_t i = 0; i < y_insp.size(); ++i)
        same = same && std::fabs(static_cast<double>(y_insp[i]) - y_atomics[i]) <= 2e-6 * absrow[i] + 1e-30;
      expect(same, "matrix_opt(csc_view) result equals the atomics path to rounding");
      if (st) {
        // the snapshot contract of matrix_opt: in-place changes reach the copy through update_values
        std::vector<T> tripled(ht.values);
        for (auto& v : tripled)
          v *= 3;
        HIP_OK(hipMemcpy(dt.values.p, tripled.data(), tripled.size() * sizeof(T), hipMemcpyHostToDevice));
        st->update_values(dt.values.p);
        std::vector<double> want3, abs3;
        host_spmv(ha, hx, 3.0, want3, abs3);
        multiply(c_opt, x, y);
        expect(close_vec(dy.host(), want3, abs3) && st->inspections() == 1,
               "matrix_opt(csc_view): update_values after an in-place change, no new inspect");
        HIP_OK(hipMemcpy(dt.values.p, ht.values.data(), ht.values.size() * sizeof(T), hipMemcpyHostToDevice));
        st->update_values(dt.values.p);
      }
    }
    // y2 = A^T x2 through transposed(a)
    std::vector<T> hx2(m);
    for (auto& v : hx2)
      v = next_val();
    dev_array<T> dx2(hx2), dy2(static_cast<std::size_t>(n));
    host_spmv(ht, hx2, 1.0, want, absrow);
    multiply(transposed(a), dx2.span(), dy2.span());
    expect(close_vec(dy2.host(), want, absrow), "multiply(transposed(a), x, y)");
    // scale(alpha, a) in place on the device, then multiply again
    scale(2.0f, a);
    host_spmv(ha, hx, 2.0, want, absrow);
    multiply(a, x, y);
    expect(close_vec(dy.host(), want, absrow), "scale(alpha, a) then multiply");
  }

  // ---- SpMM: multiply(a, B, C) on row-major mdspans ----
  {
    const I m = 3000, k = 2500, n = 24;
    const host_csr ha = random_csr(m, k, 9);
    dev_csr da(ha);
    std::vector<T> hb(static_cast<std::size_t>(k) * n);
    for (auto& v : hb)
      v = next_val();
    dev_array<T> db(hb), dc(static_cast<std::size_t>(m) * n);
    mdspan_row_major<T, I> B(db.p, k, n), C(dc.p, m, n);
    auto a = da.view();
    std::vector<double> want(static_cast<std::size_t>(m) * n, 0.0), scale(static_cast<std::size_t>(m) * n, 0.0);
    for (I r = 0; r < m; ++r)
      for (O p = ha.rowptr[r]; p < ha.rowptr[r + 1]; ++p)
        for (I j = 0; j < n; ++j) {
          const double t = static_cast<double>(ha.values[p]) * hb[static_cast<std::size_t>(ha.colind[p]) * n + j];
          want[static_cast<std::size_t>(r) * n + j] += t;
          scale[static_cast<std::size_t>(r) * n + j] += std::fabs(t);
        }
    multiply(a, B, C);
    expect(close_vec(dc.host(), want, scale), "multiply(a, B, C)");
    HIP_OK(hipMemset(dc.p, 0xFF, dc.n * sizeof(T)));
    operation_info_t info = multiply_inspect(a, B, C);
    multiply(info, a, B, C);
    expect(close_vec(dc.host(), want, scale), "multiply(info, a, B, C) after multiply_inspect");
    {
      // column-major dense operands (mdspan_col_major, detail/mdspan.hpp:31-36; test/gtest/mdspan_overlays.cpp:38-45): B
      // and C stored column by column, and the mixed pair B column-major / C row-major
      std::vector<T> hb_cm(hb.size());
      for (I r = 0; r < k; ++r)
        for (I j = 0; j < n; ++j)
          hb_cm[static_cast<std::size_t>(j) * k + r] = hb[static_cast<std::size_t>(r) * n + j];
      dev_array<T> db_cm(hb_cm), dc_cm(static_cast<std::size_t>(m) * n);
      mdspan_col_major<T, I> Bc(db_cm.p, k, n), Cc(dc_cm.p, m, n);
      HIP_OK(hipMemset(dc_cm.p, 0xFF, dc_cm.n * sizeof(T)));
      multiply(a, Bc, Cc);
      const std::vector<T> got_cm = dc_cm.host();
      std::vector<T> got_rm(got_cm.size());
      for (I r = 0; r < m; ++r)
        for (I j = 0; j < n; ++j)
          got_rm[static_cast<std::size_t>(r) * n + j] = got_cm[static_cast<std::size_t>(j) * m + r];
      expect(close_vec(got_rm, want, scale), "multiply(a, B, C) with mdspan_col_major B and C");
      HIP_OK(hipMemset(dc.p, 0xFF, dc.n * sizeof(T)));
      multiply(info, a, Bc, C);
      expect(close_vec(dc.host(), want, scale), "multiply(info, a, B col-major, C row-major)");
    }
  }

  // ---- 64-bit row offsets and fp64 values: csr_view<double, int32, int64> (views/csr_view.hpp is templated on all three;
  //      the library-wide offset_t of this backend is 32-bit, a caller's view need not be) ----
  {
    using O64 = std::int64_t;
    const I m = 9000, n = 11000, nc = 5;
    const host_csr ha = random_csr(m, n, 7);
    std::vector<O64> rp64(ha.rowptr.begin(), ha.rowptr.end());
    std::vector<double> v64(ha.values.begin(), ha.values.end());
    dev_array<double> dv(v64);
    dev_array<O64> drp(rp64);
    dev_array<I> dci(ha.colind);
    csr_view<double, I, O64> a(dv.p, drp.p, dci.p, {m, n}, static_cast<O64>(ha.nnz()));
    std::vector<T> hx32(n);
    for (auto& v : hx32)
      v = next_val();
    std::vector<double> hx(hx32.begin(), hx32.end());
    dev_array<double> dx(hx), dy(static_cast<std::size_t>(m));
    std::vector<double> want, absrow;
    host_spmv(ha, hx32, 1.0, want, absrow);
    auto close64 = [&](const std::vector<double>& got, const std::vector<double>& w, const std::vector<double>& sc) {
      bool ok = got.size() == w.size();
      for (std::size_t i = 0; ok && i < got.size(); ++i)
        ok = std::fabs(got[i] - w[i]) <= 1e-12 * sc[i] + 1e-300;
      return ok;
    };
    multiply(a, dx.span(), dy.span());
    expect(close64(dy.host(), want, absrow), "multiply(csr_view<double, int32, int64>, x, y)");
    HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(double)));
    operation_info_t info = multiply_inspect(a, dx.span(), dy.span());
    multiply(info, a, dx.span(), dy.span());
    expect(close64(dy.host(), want, absrow), "multiply(info, csr_view<double, int32, int64>, x, y)");
    HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(double)));
    matrix_opt a_opt(a);
    multiply_inspect(a_opt, dx.span(), dy.span());
    multiply(a_opt, dx.span(), dy.span());
    expect(close64(dy.host(), want, absrow), "multiply(matrix_opt(csr_view<double, int32, int64>), x, y)");
    std::vector<double> hb(static_cast<std::size_t>(n) * nc);
    for (auto& v : hb)
      v = next_val();
    dev_array<double> db(hb), dc(static_cast<std::size_t>(m) * nc);
    mdspan_row_major<double, I> B(db.p, n, nc), C(dc.p, m, nc);
    std::vector<double> wantc(static_cast<std::size_t>(m) * nc, 0.0), scalec(static_cast<std::size_t>(m) * nc, 0.0);
    for (I r = 0; r < m; ++r)
      for (O p = ha.rowptr[r]; p < ha.rowptr[r + 1]; ++p)
        for (I j = 0; j < nc; ++j) {
          const double t = static_cast<double>(ha.values[p]) * hb[static_cast<std::size_t>(ha.colind[p]) * nc + j];
          wantc[static_cast<std::size_t>(r) * nc + j] += t;
          scalec[static_cast<std::size_t>(r) * nc + j] += std::fabs(t);
        }
    multiply(a, B, C);
    expect(close64(dc.host(), wantc, scalec), "multiply(csr_view<double, int32, int64>, B, C)");
    HIP_OK(hipMemset(dc.p, 0xFF, dc.n * sizeof(double)));
    operation_info_t infoc = multiply_inspect(a, B, C);
    multiply(infoc, a, B, C);
    expect(close64(dc.host(), wantc, scalec), "multiply(info, csr_view<double, int32, int64>, B, C)");
  }

  // ---- 64-bit COLUMN indices: csr_view<float, int64, int32> / <float, int64, int64> -- the rocSPARSE slot admits them
  //      (vendor/rocsparse/types.hpp:16-24); narrowed once per index array on the device, range-checked ----
  {
    using I64 = std::int64_t;
    const I m = 7000, n = 9000;
    const host_csr ha = random_csr(m, n, 6);
    std::vector<I64> ci64(ha.colind.begin(), ha.colind.end());
    std::vector<I64> rp64(ha.rowptr.begin(), ha.rowptr.end());
    dev_array<T> dv(ha.values);
    dev_array<O> drp(ha.rowptr);
    dev_array<I64> drp64(rp64), dci(ci64);
    std::vector<T> hx(n);
    for (auto& v : hx)
      v = next_val();
    dev_array<T> dx(hx), dy(static_cast<std::size_t>(m));
    std::vector<double> want, absrow;
    host_spmv(ha, hx, 1.0, want, absrow);
    {
      csr_view<T, I64, O> a(dv.p, drp.p, dci.p, {static_cast<I64>(m), static_cast<I64>(n)}, static_cast<O>(ha.nnz()));
      multiply(a, dx.span(), dy.span());
      expect(close_vec(dy.host(), want, absrow), "multiply(csr_view<float, int64, int32>, x, y)");
      HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
      operation_info_t info = multiply_inspect(a, dx.span(), dy.span());
      multiply(info, a, dx.span(), dy.span());
      multiply(info, a, dx.span(), dy.span());
      expect(close_vec(dy.host(), want, absrow), "multiply(info, csr_view<float, int64, int32>, x, y)");
    }
    {
      csr_view<T, I64, I64> a(dv.p, drp64.p, dci.p, {static_cast<I64>(m), static_cast<I64>(n)}, static_cast<I64>(ha.nnz()));
      HIP_OK(hipMemset(dy.p, 0xFF, m * sizeof(T)));
      operation_info_t info = multiply_inspect(a, dx.span(), dy.span());
      std::vector<double> want2, abs2;
      host_spmv(ha, hx, -2.0, want2, abs2);
      multiply(info, scaled(-2.0f, a), dx.span(), dy.span());
      expect(close_vec(dy.host(), want2, abs2), "multiply(info, scaled(csr_view<float, int64, int64>), x, y)");
    }
    {
      // SpMM with 64-bit column indices: plain, inspected (twice through the plan), both offset types
      const I nc = 20;
      std::vector<T> hb(static_cast<std::size_t>(n) * nc);
      for (auto& v : hb)
        v = next_val();
      dev_array<T> db(hb), dc(static_cast<std::size_t>(m) * nc);
      mdspan_row_major<T, I> B(db.p, n, nc), C(dc.p, m, nc);
      std::vector<double> wantc(static_cast<std::size_t>(m) * nc, 0.0), scalec(static_cast<std::size_t>(m) * nc, 0.0);
      for (I r = 0; r < m; ++r)
        for (O p = ha.rowptr[r]; p < ha.rowptr[r + 1]; ++p)
          for (I j = 0; j < nc; ++j) {
            const double t = static_cast<double>(ha.values[p]) * hb[static_cast<std::size_t>(ha.colind[p]) * nc + j];
            wantc[static_cast<std::size_t>(r) * nc + j] += t;
            scalec[static_cast<std::size_t>(r) * nc + j] += std::fabs(t);
          }
      csr_view<T, I64, O> a(dv.p, drp.p, dci.p, {static_cast<I64>(m), static_cast<I64>(n)}, static_cast<O>(ha.nnz()));
      HIP_OK(hipMemset(dc.p, 0xFF, dc.n * sizeof(T)));
      multiply(a, B, C);
      expect(close_vec(dc.host(), wantc, scalec), "multiply(csr_view<float, int64, int32>, B, C)");
      csr_view<T, I64, I64> a64(dv.p, drp64.p, dci.p, {static_cast<I64>(m), static_cast<I64>(n)}, static_cast<I64>(ha.nnz()));
      HIP_OK(hipMemset(dc.p, 0xFF, dc.n * sizeof(T)));
      operation_info_t info = multiply_inspect(a64, B, C);
      multiply(info, a64, B, C);
      multiply(info, a64, B, C);
      expect(close_vec(dc.host(), wantc, scalec), "multiply(info, csr_view<float, int64, int64>, B, C)");
    }
    {
      std::vector<I64> bad = ci64;
      bad[bad.size() / 2] = (static_cast<I64>(1) << 32) + 5;  // would wrap to column 5
      dev_array<I64> dbad(bad);
      csr_view<T, I64, O> a(dv.p, drp.p, dbad.p, {static_cast<I64>(m), static_cast<I64>(n)}, static_cast<O>(ha.nnz()));
      bool threw = false;
      try {
        multiply(a, dx.span(), dy.span());
      } catch (const std::invalid_argument&) {
        threw = true;
      }
      expect(threw, "a 64-bit column index outside the matrix throws std::invalid_argument");
    }
  }

  // ---- SpGEMM: multiply_compute / multiply_fill, then the symbolic / numeric split with reuse ----
  {
    const I m = 4000, k = 3000, n = 3500;
    const host_csr ha = random_csr(m, k, 6), hb = random_csr(k, n, 5);
    dev_csr da(ha), db(hb);
    // host result (ordered map per row: ascending columns, like the backend's output)
    host_csr hc;
    hc.m = m;
    hc.n = n;
    hc.rowptr.assign(static_cast<std::size_t>(m) + 1, 0);
    std::vector<double> hc_vals, hc_abs;
    for (I r = 0; r < m; ++r) {
      std::map<I, std::pair<double, double>> acc;
      for (O p = ha.rowptr[r]; p < ha.rowptr[r + 1]; ++p)
        for (O q = hb.rowptr[ha.colind[p]]; q < hb.rowptr[ha.colind[p] + 1]; ++q) {
          const double t = static_cast<double>(ha.values[p]) * hb.values[q];
          acc[hb.colind[q]].first += t;
          acc[hb.colind[q]].second += std::fabs(t);
        }
      for (auto& [c, v] : acc) {
        hc.colind.push_back(c);
        hc_vals.push_back(v.first);
        hc_abs.push_back(v.second);
      }
      hc.rowptr[static_cast<std::size_t>(r) + 1] = static_cast<O>(hc.colind.size());
    }
    dev_array<O> c_rowptr(static_cast<std::size_t>(m) + 1);
    csr_view<T, I, O> c(static_cast<T*>(nullptr), c_rowptr.p, static_cast<I*>(nullptr), {m, n}, 0);
    auto a = da.view(), b = db.view();
    operation_info_t info = multiply_compute(a, b, c);
    expect(static_cast<O>(info.result_nnz()) == hc.nnz(), "multiply_compute: nnz(C)");
    dev_array<T> c_values(static_cast<std::size_t>(info.result_nnz()));
    dev_array<I> c_colind(static_cast<std::size_t>(info.result_nnz()));
    c.update(c_values.span(), std::span<O>(c_rowptr.p, c_rowptr.n), c_colind.span());
    multiply_fill(info, a, b, c);
    expect(c_rowptr.host() == hc.rowptr, "multiply_fill: rowptr(C)");
    expect(c_colind.host() == hc.colind, "multiply_fill: colind(C) ascending and exact");
    expect(close_vec(c_values.host(), hc_vals, hc_abs), "multiply_fill: values(C)");
    // reuse: numeric phase twice with rescaled A (multiply_spgemm.hpp:178-214 call shape)
    spgemm_state_t state;
    multiply_symbolic_compute(state, a, b, c);
    multiply_symbolic_fill(state, a, b, c);
    for (float alpha : {2.0f, -0.5f, 4.0f}) {
      HIP_OK(hipMemset(c_values.p, 0xFF, c_values.n * sizeof(T)));
      multiply_numeric(state, scaled(alpha, a), b, c);
      std::vector<double> want(hc_vals), sc(hc_abs);
      for (std::size_t i = 0; i < want.size(); ++i) {
        want[i] *= alpha;
        sc[i] *= std::fabs(alpha);
      }
      expect(close_vec(c_values.host(), want, sc), "multiply_numeric(state, scaled(alpha, a), b, c)");
      expect(c_colind.host() == hc.colind, "multiply_numeric keeps colind(C)");
    }
  }

  // ---- add(a, b, c) and transpose(a, b) ----
  {
    const I m = 5000, n = 4000;
    const host_csr ha = random_csr(m, n, 7), hb = random_csr(m, n, 5);
    dev_csr da(ha), db(hb);
    host_csr hc;
    hc.m = m;
    hc.n = n;
    hc.rowptr.assign(static_cast<std::size_t>(m) + 1, 0);
    std::vector<double> hv, habs;
    for (I r = 0; r < m; ++r) {
      std::map<I, double> acc;
      for (O p = ha.rowptr[r]; p < ha.rowptr[r + 1]; ++p)
        acc[ha.colind[p]] += ha.values[p];
      for (O p = hb.rowptr[r]; p < hb.rowptr[r + 1]; ++p)
        acc[hb.colind[p]] += hb.values[p];
      for (auto& [cc, v] : acc) {
        hc.colind.push_back(cc);
        hv.push_back(v);
        habs.push_back(std::fabs(v) + 1.0);
      }
      hc.rowptr[static_cast<std::size_t>(r) + 1] = static_cast<O>(hc.colind.size());
    }
    dev_array<O> c_rowptr(static_cast<std::size_t>(m) + 1);
    csr_view<T, I, O> c(static_cast<T*>(nullptr), c_rowptr.p, static_cast<I*>(nullptr), {m, n}, 0);
    auto a = da.view(), b = db.view();
    operation_info_t info = add_inspect(a, b, c);
    expect(static_cast<O>(info.result_nnz()) == hc.nnz(), "add_inspect: nnz(C)");
    dev_array<T> c_values(static_cast<std::size_t>(info.result_nnz()));
    dev_array<I> c_colind(static_cast<std::size_t>(info.result_nnz()));
    c.update(c_values.span(), std::span<O>(c_rowptr.p, c_rowptr.n), c_colind.span());
    add_compute(info, a, b, c);
    expect(c_rowptr.host() == hc.rowptr && c_colind.host() == hc.colind, "add_compute: structure of C");
    expect(close_vec(c_values.host(), hv, habs), "add_compute: values of C");

    const host_csr ht = host_transpose(ha);
    dev_array<T> t_values(static_cast<std::size_t>(ha.nnz()));
    dev_array<O> t_rowptr(static_cast<std::size_t>(n) + 1);
    dev_array<I> t_colind(static_cast<std::size_t>(ha.nnz()));
    csr_view<T, I, O> t(t_values.p, t_rowptr.p, t_colind.p, {n, m}, ha.nnz());
    transpose(a, t);
    expect(t_rowptr.host() == ht.rowptr && t_colind.host() == ht.colind && t_values.host() == ht.values,
           "transpose(a, b): bit-identical to the counting sort");
  }

  // ---- triangular_solve: lower / explicit diagonal with inspect, upper / implicit unit diagonal without ----
  {
    const I m = 6000;
    host_csr hl;
    hl.m = hl.n = m;
    hl.rowptr.assign(static_cast<std::size_t>(m) + 1, 0);
    for (I r = 0; r < m; ++r) {
      std::vector<I> cols;
      for (int t = 0; t < 4 && r > 0; ++t) {
        const I cc = static_cast<I>(next_u32() % static_cast<std::uint32_t>(r));
        if (std::find(cols.begin(), cols.end(), cc) == cols.end())
          cols.push_back(cc);
      }
      std::sort(cols.begin(), cols.end());
      for (I cc : cols) {
        hl.colind.push_back(cc);
        hl.values.push_back(next_val() * 0.05f);
      }
      hl.colind.push_back(r);
      hl.values.push_back(2.0f + 0.25f * static_cast<T>(next_u32() % 5));
      hl.rowptr[static_cast<std::size_t>(r) + 1] = static_cast<O>(hl.colind.size());
    }
    std::vector<T> hb(m);
    for (auto& v : hb)
      v = next_val();
    std::vector<double> want(m), sc(m);
    for (I r = 0; r < m; ++r) {
      double s = hb[r], ab = std::fabs(hb[r]), d = 1.0;
      for (O p = hl.rowptr[r]; p < hl.rowptr[r + 1]; ++p) {
        if (hl.colind[p] < r) {
          s -= hl.values[p] * want[hl.colind[p]];
          ab += std::fabs(hl.values[p] * want[hl.colind[p]]);
        } else if (hl.colind[p] == r)
          d = hl.values[p];
      }
      want[r] = s / d;
      sc[r] = 100.0 * ab / std::fabs(d);
    }
    dev_csr dl(hl);
    dev_array<T> db(hb), dx(static_cast<std::size_t>(m));
    auto l = dl.view();
    operation_info_t info = triangular_solve_inspect(l, lower_triangle_t{}, explicit_diagonal_t{}, db.span(), dx.span());
    triangular_solve(info, l, lower_triangle_t{}, explicit_diagonal_t{}, db.span(), dx.span());
    expect(close_vec(dx.host(), want, sc), "triangular_solve(info, L, lower, explicit, b, x)");
    {  // scaled right-hand side (examples/simple_sptrsv.cpp:49-53): x = inv(L) (3 b) = 3 inv(L) b
      std::vector<double> want3(want), sc3(sc);
      for (std::size_t i = 0; i < want3.size(); ++i) {
        want3[i] *= 3.0;
        sc3[i] *= 3.0;
      }
      triangular_solve(l, lower_triangle_t{}, explicit_diagonal_t{}, scaled(3.0f, db.span()), dx.span());
      expect(close_vec(dx.host(), want3, sc3), "triangular_solve(L, lower, explicit, scaled(3, b), x)");
    }
    // U = L^T as CSR, unit diagonal: x_i = b_i - sum_{k > i} u_ik x_k
    const host_csr hu = host_transpose(hl);
    for (I r = m - 1; r >= 0; --r) {
      double s = hb[r], ab = std::fabs(hb[r]);
      for (O p = hu.rowptr[r]; p < hu.rowptr[r + 1]; ++p)
        if (hu.colind[p] > r) {
          s -= hu.values[p] * want[hu.colind[p]];
          ab += std::fabs(hu.values[p] * want[hu.colind[p]]);
        }
      want[r] = s;
      sc[r] = 100.0 * ab;
    }
    dev_csr du(hu);
    triangular_solve(du.view(), upper_triangle_t{}, implicit_unit_diagonal_t{}, db.span(), dx.span());
    expect(close_vec(dx.host(), want, sc), "triangular_solve(U, upper, implicit unit, b, x)");
  }

  HIP_OK(hipDeviceSynchronize());
  std::printf("dropin_run: %d checks, %d failed\n", g_checks, g_failed);
  return g_failed == 0 ? 0 : 1;
}
