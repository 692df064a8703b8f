// C++ device tests of the gfx950 backend through the standalone API mirror
// (include/spblas_gfx950/spblas.hpp).  The cases restate the reference's device tests:
//   SpMV, SpMV_Ascaled, SpMV_BScaled   /root/reference/test/gtest/device/spmv_test.cpp:11-146
//   SpGEMM, SpGEMM_AScaled             test/gtest/device/spgemm_test.cpp:12-97
//   SpGEMMReuse                        test/gtest/device/spgemm_reuse_test.cpp:12-114
//   SpMM (n in {1,8,32,64,512})        test/gtest/spmm_test.cpp:6-44 (no device SpMM test exists)
// with the same shapes (util.hpp:27-29), the same inline comparator loops and the same
// EXPECT_EQ_ tolerance (util.hpp:7-23).  No gtest in this image: plain checks, exit code = failures.
// Build: see spblas-reference_amd/_build.py (g++ -std=c++20, links the C-ABI library + amdhip64).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <random>
#include <set>
#include <tuple>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <spblas_gfx950/spblas.hpp>

using value_t = float;
using index_t = spblas::index_t;
using offset_t = spblas::offset_t;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                                 \
  do {                                                                                              \
    ++g_checks;                                                                                     \
    if (!(cond)) {                                                                                  \
      ++g_fail;                                                                                     \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);                                    \
    }                                                                                               \
  } while (0)

// EXPECT_EQ_ for floating point (test/gtest/util.hpp:7-23)
static bool near_ref(value_t t, value_t u) {
  const double eps = 64.0 * std::numeric_limits<value_t>::epsilon();
  const double norm = std::min<double>(std::abs((double) t) + std::abs((double) u), std::numeric_limits<value_t>::max());
  const double abs_error = std::max<double>(std::numeric_limits<value_t>::min(), eps * norm);
  return std::abs((double) t - (double) u) <= abs_error;
}

static const std::vector<std::tuple<int, int, int>> dims = {{1000, 100, 100}, {100, 1000, 10000}, {40, 40, 1000}};

// Same distribution as spblas::generate_csr (backend/generate.hpp:49-120): nnz distinct (i,j),
// values U[0,100), column order shuffled inside each row.
struct host_csr {
  std::vector<value_t> values;
  std::vector<offset_t> rowptr;
  std::vector<index_t> colind;
  spblas::index<index_t> shape;
  offset_t nnz;
};
static host_csr generate_csr(int m, int n, int nnz, unsigned seed = 0) {
  std::mt19937 g(seed);
  std::uniform_int_distribution<int> dr(0, m - 1), dc(0, n - 1);
  std::uniform_real_distribution<value_t> dv(0, 100);
  std::set<std::pair<int, int>> entries;
  while ((int) entries.size() < nnz)
    entries.emplace(dr(g), dc(g));
  host_csr a;
  a.shape = spblas::index<index_t>(m, n);
  a.nnz = nnz;
  a.rowptr.assign(m + 1, 0);
  for (auto& e : entries) {
    a.rowptr[e.first + 1]++;
    a.colind.push_back(e.second);
    a.values.push_back(dv(g));
  }
  for (int i = 0; i < m; i++)
    a.rowptr[i + 1] += a.rowptr[i];
  for (int i = 0; i < m; i++)
    std::shuffle(a.colind.begin() + a.rowptr[i], a.colind.begin() + a.rowptr[i + 1], g);
  return a;
}

template <typename T>
struct dvec {
  T* p = nullptr;
  size_t n = 0;
  explicit dvec(size_t count) : n(count) {
    if (hipMalloc((void**) &p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess)
      std::abort();
  }
  explicit dvec(const std::vector<T>& h) : dvec(h.size()) {
    upload(h);
  }
  void upload(const std::vector<T>& h) {
    if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      std::abort();
  }
  std::vector<T> download() const {
    std::vector<T> h(n);
    if (n && hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess)
      std::abort();
    return h;
  }
  ~dvec() {
    (void) hipFree(p);
  }
  dvec(const dvec&) = delete;
};

struct device_csr {
  dvec<value_t> values;
  dvec<offset_t> rowptr;
  dvec<index_t> colind;
  spblas::csr_view<value_t, index_t, offset_t> view;
  explicit device_csr(const host_csr& h)
      : values(h.values), rowptr(h.rowptr), colind(h.colind), view(values.p, rowptr.p, colind.p, h.shape, h.nnz) {}
};

static void test_spmv() {
  for (auto&& [m, n, nnz] : dims) {
    for (int mode = 0; mode < 3; ++mode) {  // 0: plain, 1: scaled(alpha, a), 2: scaled(alpha, b)
      for (int alpha : {-10, 1, 5}) {
        if (mode == 0 && alpha != 1)
          continue;
        auto h = generate_csr(m, n, nnz);
        device_csr a(h);
        std::vector<value_t> b(n, 1), c(m, 0);
        dvec<value_t> d_b(b), d_c(c);
        std::span<value_t> b_span(d_b.p, n), c_span(d_c.p, m);
        if (mode == 0)
          spblas::multiply(a.view, b_span, c_span);  // device/spmv_test.cpp:34
        else if (mode == 1)
          spblas::multiply(spblas::scaled(alpha, a.view), b_span, c_span);  // :81
        else
          spblas::multiply(a.view, spblas::scaled(alpha, b_span), c_span);  // :128
        c = d_c.download();
        for (int i = 0; i < m; i++) {
          value_t ref = 0;
          for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
            ref += (mode == 0   ? h.values[p] * b[h.colind[p]]
                    : mode == 1 ? alpha * h.values[p] * b[h.colind[p]]
                                : h.values[p] * alpha * b[h.colind[p]]);
          CHECK(near_ref(ref, c[i]));
        }
        if (mode == 0) {  // inspect + execute gives the same answer (README.md:36-46 call shape)
          auto info = spblas::multiply_inspect(a.view, b_span, c_span);
          dvec<value_t> d_c2(std::vector<value_t>(m, -1));
          std::span<value_t> c2_span(d_c2.p, m);
          spblas::multiply(info, a.view, b_span, c2_span);
          auto c2 = d_c2.download();
          for (int i = 0; i < m; i++) {
            if (!near_ref(c[i], c2[i]))
              std::printf("  inspect/execute mismatch: dims (%d,%d,%d) row %d len %d: %g vs %g\n", m, n, nnz, i,
                          (int) (h.rowptr[i + 1] - h.rowptr[i]), c[i], c2[i]);
            CHECK(near_ref(c[i], c2[i]));
          }
        }
      }
    }
  }
  // error behaviour: shape mismatch -> std::invalid_argument (multiply_impl.hpp:37-41)
  auto h = generate_csr(40, 40, 1000);
  device_csr a(h);
  dvec<value_t> d_b(39), d_c(40);
  bool threw = false;
  try {
    spblas::multiply(a.view, std::span<value_t>(d_b.p, 39), std::span<value_t>(d_c.p, 40));
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
  threw = false;
  try {  // conjugated views are rejected (vendor/rocsparse/detail/spmv_impl.hpp:29-33)
    spblas::multiply(spblas::conjugated(a.view), std::span<value_t>(d_c.p, 40), std::span<value_t>(d_c.p, 40));
  } catch (const std::runtime_error&) {
    threw = true;
  }
  CHECK(threw);
  // matrix_opt caches the plan; CSC operand = transposed (test/gtest/spmv_test.cpp:110-208)
  spblas::matrix_opt a_opt(a.view);
  dvec<value_t> d_x(std::vector<value_t>(40, 1)), d_y(40), d_yt(40);
  std::span<value_t> xs(d_x.p, 40), ys(d_y.p, 40), yts(d_yt.p, 40);
  auto info = spblas::multiply_inspect(a_opt, xs, ys);
  spblas::multiply(a_opt, xs, ys);
  spblas::multiply(spblas::transposed(a.view), xs, yts);
  auto y = d_y.download(), yt = d_yt.download();
  std::vector<value_t> ref(40, 0), reft(40, 0);
  for (int i = 0; i < 40; i++)
    for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++) {
      ref[i] += h.values[p];
      reft[h.colind[p]] += h.values[p];
    }
  for (int i = 0; i < 40; i++) {
    CHECK(near_ref(ref[i], y[i]));
    CHECK(near_ref(reft[i], yt[i]));
  }
}

static void test_spmm() {
  for (auto&& [m, k, nnz] : dims) {
    for (int n : {1, 8, 32, 64, 512}) {
      auto h = generate_csr(m, k, nnz);
      device_csr a(h);
      std::mt19937 g(0);
      std::uniform_real_distribution<value_t> d(0, 100);
      std::vector<value_t> b((size_t) k * n), c((size_t) m * n, 0);
      for (auto& v : b)
        v = d(g);
      dvec<value_t> d_b(b), d_c(c);
      spblas::mdspan_row_major<value_t, index_t> bm(d_b.p, k, n), cm(d_c.p, m, n);
      auto info = spblas::multiply_inspect(a.view, bm, cm);  // examples/spmm_csr.cpp:45-46
      spblas::multiply(info, a.view, bm, cm);
      c = d_c.download();
      std::vector<value_t> ref((size_t) m * n, 0);
      for (int i = 0; i < m; i++)
        for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
          for (int j = 0; j < n; j++)
            ref[(size_t) i * n + j] += h.values[p] * b[(size_t) h.colind[p] * n + j];
      bool ok = true;
      for (size_t i = 0; i < ref.size(); i++)
        ok &= near_ref(ref[i], c[i]);
      CHECK(ok);
    }
  }
}

// per-row comparison of test/gtest/device/spgemm_test.cpp:56-93
static void check_spgemm(const host_csr& a, const host_csr& b, value_t alpha, int m, const std::vector<value_t>& cv,
                         const std::vector<offset_t>& cr, const std::vector<index_t>& cc) {
  for (int i = 0; i < m; i++) {
    std::map<index_t, value_t> ref, acc;
    for (auto p = a.rowptr[i]; p < a.rowptr[i + 1]; p++)
      for (auto q = b.rowptr[a.colind[p]]; q < b.rowptr[a.colind[p] + 1]; q++)
        ref[b.colind[q]] += alpha * a.values[p] * b.values[q];
    for (auto p = cr[i]; p < cr[i + 1]; p++)
      acc[cc[p]] += cv[p];
    bool ok = ref.size() == acc.size();  // distinct-column count must match exactly (:93)
    for (auto p = cr[i]; p + 1 < cr[i + 1]; p++)
      ok &= cc[p] < cc[p + 1];  // ascending like spgemm_gustavsons.hpp:42
    for (auto& [j, v] : acc)
      ok &= ref.count(j) && near_ref(ref[j], v);
    CHECK(ok);
  }
}

static void test_spgemm() {
  for (auto&& [m, k, nnz] : dims) {
    for (int n : {m, k}) {
      for (int variant = 0; variant < 3; ++variant) {  // 0: state API, 1: A scaled by 2, 2: operation_info_t API
        auto ha = generate_csr(m, k, nnz), hb = generate_csr(k, n, nnz, 1);
        device_csr a(ha), b(hb);
        dvec<offset_t> d_c_rowptr(m + 1);
        spblas::csr_view<value_t, index_t, offset_t> d_c(nullptr, d_c_rowptr.p, nullptr, {m, n}, 0);
        const value_t alpha = variant == 1 ? 2.0f : 1.0f;
        spblas::spgemm_state_t state;
        spblas::operation_info_t info;
        std::int64_t cn;
        if (variant == 2) {
          info = spblas::multiply_compute(a.view, b.view, d_c);  // examples/simple_spgemm.cpp:52
          cn = info.result_nnz();
        } else if (variant == 1) {
          spblas::multiply_compute(state, spblas::scaled(alpha, a.view), b.view, d_c);
          cn = state.result_nnz();
        } else {
          spblas::multiply_compute(state, a.view, b.view, d_c);  // device/spgemm_test.cpp:42-43
          cn = state.result_nnz();
        }
        dvec<value_t> d_c_values(cn);
        dvec<index_t> d_c_colind(cn);
        d_c.update(std::span<value_t>(d_c_values.p, cn), std::span<offset_t>(d_c_rowptr.p, m + 1),
                   std::span<index_t>(d_c_colind.p, cn), {m, n}, (offset_t) cn);
        if (variant == 2)
          spblas::multiply_fill(info, a.view, b.view, d_c);
        else if (variant == 1)
          spblas::multiply_fill(state, spblas::scaled(alpha, a.view), b.view, d_c);
        else
          spblas::multiply_fill(state, a.view, b.view, d_c);
        check_spgemm(ha, hb, alpha, m, d_c_values.download(), d_c_rowptr.download(), d_c_colind.download());
      }
    }
  }
  // reuse: symbolic once, numeric three times with new values (spgemm_reuse_test.cpp:42-70)
  auto [m, k, nnz] = dims[1];
  auto ha = generate_csr(m, k, nnz), hb = generate_csr(k, m, nnz, 1);
  device_csr a(ha), b(hb);
  dvec<offset_t> d_c_rowptr(m + 1);
  spblas::csr_view<value_t, index_t, offset_t> d_c(nullptr, d_c_rowptr.p, nullptr, {m, m}, 0);
  spblas::spgemm_state_t state;
  spblas::multiply_symbolic_compute(state, a.view, b.view, d_c);
  const auto cn = state.result_nnz();
  dvec<value_t> d_c_values(cn);
  dvec<index_t> d_c_colind(cn);
  d_c.update(std::span<value_t>(d_c_values.p, cn), std::span<offset_t>(d_c_rowptr.p, m + 1),
             std::span<index_t>(d_c_colind.p, cn), {m, m}, (offset_t) cn);
  spblas::multiply_symbolic_fill(state, a.view, b.view, d_c);
  std::mt19937 g(0);
  for (int it = 0; it < 3; it++) {
    if (it) {
      std::uniform_real_distribution<value_t> d(0, 100);
      for (auto& v : ha.values)
        v = d(g);
      for (auto& v : hb.values)
        v = d(g);
      a.values.upload(ha.values);
      b.values.upload(hb.values);
    }
    spblas::multiply_numeric(state, a.view, b.view, d_c);
    check_spgemm(ha, hb, 1.0f, m, d_c_values.download(), d_c_rowptr.download(), d_c_colind.download());
  }
  // out of memory (spgemm_gustavsons.hpp:44-48)
  spblas::csr_view<value_t, index_t, offset_t> small(d_c_values.p, d_c_rowptr.p, d_c_colind.p, {m, m},
                                                     (offset_t) cn - 1);
  bool threw = false;
  try {
    spblas::multiply_fill(state, a.view, b.view, small);
  } catch (const std::runtime_error&) {
    threw = true;
  }
  CHECK(threw);
}

// test/gtest/device/rocsparse/spgemm_4args_test.cpp:11-110 (+ _AScaled/_BScaled/_DScaled): C = alpha*A*B + beta*D
static void test_spgemm_4args() {
  for (auto&& [m, k, nnz] : dims) {
    for (int n : {m, k}) {
      for (int variant = 0; variant < 4; ++variant) {  // 0 plain, 1 A scaled, 2 B scaled, 3 D scaled
        auto ha = generate_csr(m, k, nnz), hb = generate_csr(k, n, nnz, 1), hd = generate_csr(m, n, nnz, 2);
        device_csr a(ha), b(hb), d(hd);
        dvec<offset_t> d_c_rowptr(m + 1);
        spblas::csr_view<value_t, index_t, offset_t> d_c(nullptr, d_c_rowptr.p, nullptr, {m, n}, 0);
        const value_t sa = variant == 1 ? 2.0f : 1.0f, sb = variant == 2 ? 2.0f : 1.0f, sd = variant == 3 ? 2.0f : 1.0f;
        auto A = spblas::scaled(sa, a.view);
        auto B = spblas::scaled(sb, b.view);
        auto D = spblas::scaled(sd, d.view);
        spblas::spgemm_state_t state;
        spblas::multiply_compute(state, A, B, d_c, D);  // :55
        const auto cn = state.result_nnz();
        dvec<value_t> d_c_values(cn);
        dvec<index_t> d_c_colind(cn);
        d_c.update(std::span<value_t>(d_c_values.p, cn), std::span<offset_t>(d_c_rowptr.p, m + 1),
                   std::span<index_t>(d_c_colind.p, cn), {m, n}, (offset_t) cn);
        spblas::multiply_fill(state, A, B, d_c, D);  // :65
        auto cv = d_c_values.download();
        auto cr = d_c_rowptr.download();
        auto cc = d_c_colind.download();
        for (int i = 0; i < m; i++) {  // :78-108
          std::map<index_t, value_t> ref, acc;
          for (auto p = ha.rowptr[i]; p < ha.rowptr[i + 1]; p++)
            for (auto q = hb.rowptr[ha.colind[p]]; q < hb.rowptr[ha.colind[p] + 1]; q++)
              ref[hb.colind[q]] += sa * sb * ha.values[p] * hb.values[q];
          for (auto p = hd.rowptr[i]; p < hd.rowptr[i + 1]; p++)
            ref[hd.colind[p]] += sd * hd.values[p];
          for (auto p = cr[i]; p < cr[i + 1]; p++)
            acc[cc[p]] += cv[p];
          bool ok = ref.size() == acc.size() && (std::int64_t) acc.size() == cr[i + 1] - cr[i];
          for (auto& [j, v] : acc)
            ok &= ref.count(j) && near_ref(ref[j], v);
          CHECK(ok);
        }
        CHECK(cr[m] == cn);
      }
    }
  }
}

// test/gtest/add_test.cpp:9-60: add_inspect -> allocate -> update -> add_compute
static void test_add() {
  for (auto&& [m, n, nnz] : dims) {
    for (int variant = 0; variant < 2; ++variant) {  // 0 plain, 1 scaled(2, a) + scaled(-0.5, b)
      auto ha = generate_csr(m, n, nnz), hb = generate_csr(m, n, nnz, 1);
      device_csr a(ha), b(hb);
      dvec<offset_t> d_c_rowptr(m + 1);
      spblas::csr_view<value_t, index_t, offset_t> d_c(nullptr, d_c_rowptr.p, nullptr, {m, n}, 0);
      const value_t sa = variant ? 2.0f : 1.0f, sb = variant ? -0.5f : 1.0f;
      auto A = spblas::scaled(sa, a.view);
      auto B = spblas::scaled(sb, b.view);
      auto info = spblas::add_inspect(A, B, d_c);  // :28
      const auto cn = info.result_nnz();
      dvec<value_t> d_c_values(cn);
      dvec<index_t> d_c_colind(cn);
      d_c.update(std::span<value_t>(d_c_values.p, cn), std::span<offset_t>(d_c_rowptr.p, m + 1),
                 std::span<index_t>(d_c_colind.p, cn), {m, n}, (offset_t) cn);
      spblas::add_compute(info, A, B, d_c);  // :35
      auto cv = d_c_values.download();
      auto cr = d_c_rowptr.download();
      auto cc = d_c_colind.download();
      for (int i = 0; i < m; i++) {  // :40-57
        std::map<index_t, value_t> ref;
        for (auto p = ha.rowptr[i]; p < ha.rowptr[i + 1]; p++)
          ref[ha.colind[p]] += sa * ha.values[p];
        for (auto p = hb.rowptr[i]; p < hb.rowptr[i + 1]; p++)
          ref[hb.colind[p]] += sb * hb.values[p];
        bool ok = (std::int64_t) ref.size() == cr[i + 1] - cr[i];
        auto it = ref.begin();
        for (auto p = cr[i]; ok && p < cr[i + 1]; p++, ++it)
          ok &= it->first == cc[p] && near_ref(it->second, cv[p]);  // ascending columns, SPA values
        CHECK(ok);
      }
    }
  }
  // shape mismatch (add_impl.hpp:44-47) and too little room (:67-72)
  auto ha = generate_csr(40, 30, 100), hb = generate_csr(40, 31, 100, 1), hc = generate_csr(40, 30, 100, 2);
  device_csr a(ha), b(hb), c2(hc);
  dvec<offset_t> rp(41);
  spblas::csr_view<value_t, index_t, offset_t> d_c(nullptr, rp.p, nullptr, {40, 30}, 0);
  bool threw = false;
  try {
    spblas::add_inspect(a.view, b.view, d_c);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
  dvec<value_t> v(3);
  dvec<index_t> ci(3);
  spblas::csr_view<value_t, index_t, offset_t> small(v.p, rp.p, ci.p, {40, 30}, 3);
  threw = false;
  try {
    spblas::add(a.view, c2.view, small);
  } catch (const std::runtime_error&) {
    threw = true;
  }
  CHECK(threw);
}

// test/gtest/triangular_solve_test.cpp:6-104: device triangular_solve against the test's own
// reference loop; beyond the reference's b = 0 / unit-diagonal case also explicit diagonals.
template <typename Triangle, typename Diag>
static void trsv_case(int n, int nnz, Triangle t, Diag d, bool zero_b) {
  constexpr bool upper = std::is_same_v<Triangle, spblas::upper_triangle_t>;
  constexpr bool unit = std::is_same_v<Diag, spblas::implicit_unit_diagonal_t>;
  auto h = generate_csr(n, n, nnz);
  for (auto& v : h.values)
    v *= 1e-3f;  // :72-74
  if (!unit) {   // make sure every row stores a dominant diagonal
    host_csr g;
    g.shape = h.shape;
    g.rowptr.assign(n + 1, 0);
    for (int i = 0; i < n; i++) {
      for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
        if (h.colind[p] != i) {
          g.colind.push_back(h.colind[p]);
          g.values.push_back(h.values[p]);
        }
      g.colind.push_back(i);
      g.values.push_back(2.0f + 0.001f * i);
      g.rowptr[i + 1] = (offset_t) g.colind.size();
    }
    g.nnz = (offset_t) g.colind.size();
    h = g;
  }
  device_csr a(h);
  std::vector<value_t> b(n, 0.0f), x_ref(n, 0.0f);
  if (!zero_b)
    for (int i = 0; i < n; i++)
      b[i] = 1.0f + 0.01f * (i % 17);
  dvec<value_t> d_b(b), d_x(std::vector<value_t>(n, 1.0f));  // x starts at 1 like :69
  std::span<value_t> bs(d_b.p, n), xs(d_x.p, n);
  auto info = spblas::triangular_solve_inspect(a.view, t, d, bs, xs);
  spblas::triangular_solve(info, a.view, t, d, bs, xs);
  for (int s = 0; s < n; s++) {  // reference_triangular_solve, :17-58
    const int row = upper ? n - 1 - s : s;
    value_t tmp = b[row], diag_val = 0;
    for (auto j = h.rowptr[row]; j < h.rowptr[row + 1]; j++) {
      const int col = h.colind[j];
      if (upper ? col > row : col < row)
        tmp -= h.values[j] * x_ref[col];
      else if (col == row)
        diag_val = h.values[j];
    }
    x_ref[row] = unit ? tmp : tmp / diag_val;
  }
  auto x = d_x.download();
  bool ok = true;
  for (int i = 0; i < n; i++)
    ok &= near_ref(x_ref[i], x[i]);
  CHECK(ok);
  // one-shot form without inspect
  dvec<value_t> d_x2(std::vector<value_t>(n, 1.0f));
  std::span<value_t> xs2(d_x2.p, n);
  spblas::triangular_solve(a.view, t, d, bs, xs2);
  CHECK(d_x2.download() == x);
  // scaled right-hand side (examples/simple_sptrsv.cpp:49-53): the solve is linear in b
  dvec<value_t> d_x3(std::vector<value_t>(n, 1.0f));
  std::span<value_t> xs3(d_x3.p, n);
  spblas::triangular_solve(a.view, t, d, spblas::scaled(2.0f, bs), xs3);
  auto x3 = d_x3.download();
  bool ok3 = true;
  for (int i = 0; i < n; i++)
    ok3 &= near_ref(2.0f * x_ref[i], x3[i]);
  CHECK(ok3);
}

static void test_triangular_solve() {
  const std::vector<std::tuple<int, int, int>> square_dims = {{1000, 1000, 100}, {100, 100, 100}, {40, 40, 1000}};
  for (auto&& [m, n, nnz] : square_dims) {
    (void) m;
    trsv_case(n, nnz, spblas::lower_triangle, spblas::implicit_unit_diagonal, true);   // :88-94
    trsv_case(n, nnz, spblas::upper_triangle, spblas::implicit_unit_diagonal, true);   // :96-102
    trsv_case(n, nnz, spblas::lower_triangle, spblas::implicit_unit_diagonal, false);
    trsv_case(n, nnz, spblas::upper_triangle, spblas::implicit_unit_diagonal, false);
    trsv_case(n, nnz, spblas::lower_triangle, spblas::explicit_diagonal, false);
    trsv_case(n, nnz, spblas::upper_triangle, spblas::explicit_diagonal, false);
  }
  bool threw = false;
  try {
    auto h = generate_csr(30, 20, 50);
    device_csr a(h);
    dvec<value_t> b(30), x(20);
    spblas::triangular_solve(a.view, spblas::lower_triangle, spblas::explicit_diagonal, std::span<value_t>(b.p, 30),
                             std::span<value_t>(x.p, 20));
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
}

// test/gtest/transpose_test.cpp:9-71: B = A^T, then (row, col, value) triples must agree
static void test_transpose() {
  for (auto&& [m, k, nnz] : dims) {
    auto h = generate_csr(m, k, nnz);
    device_csr a(h);
    dvec<offset_t> d_rp(k + 1);
    dvec<index_t> d_ci(nnz);
    dvec<value_t> d_v(nnz);
    spblas::csr_view<value_t, index_t, offset_t> b(d_v.p, d_rp.p, d_ci.p, {k, m}, nnz);
    auto info = spblas::transpose_inspect(a.view, b);
    spblas::transpose(info, a.view, b);
    auto rp = d_rp.download();
    auto ci = d_ci.download();
    auto v = d_v.download();
    std::vector<std::tuple<int, int, value_t>> ref, got;
    for (int i = 0; i < m; i++)
      for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
        ref.emplace_back(h.colind[p], i, h.values[p]);
    for (int j = 0; j < k; j++)
      for (auto p = rp[j]; p < rp[j + 1]; p++)
        got.emplace_back(j, ci[p], v[p]);
    std::sort(ref.begin(), ref.end());
    bool sorted_rows = true;  // counting sort leaves every output row in source (ascending row) order
    for (size_t t = 1; t < got.size(); t++)
      sorted_rows &= !(std::get<0>(got[t - 1]) == std::get<0>(got[t]) && std::get<1>(got[t - 1]) > std::get<1>(got[t]));
    CHECK(sorted_rows);
    if (!(ref == got)) {
      std::printf("  transpose mismatch dims (%d,%d,%d): sizes %zu %zu\n", m, k, nnz, ref.size(), got.size());
      for (size_t t = 0, shown = 0; t < std::min(ref.size(), got.size()) && shown < 5; t++)
        if (ref[t] != got[t]) {
          std::printf("    [%zu] ref (%d,%d,%g) got (%d,%d,%g)\n", t, std::get<0>(ref[t]), std::get<1>(ref[t]), std::get<2>(ref[t]),
                      std::get<0>(got[t]), std::get<1>(got[t]), std::get<2>(got[t]));
          ++shown;
        }
    }
    CHECK(ref == got);
  }
}

// scale(alpha, t) (algorithms/scale_impl.hpp:13-31): one IEEE multiply per stored value, in place
static void test_scale() {
  for (auto&& [m, k, nnz] : dims) {
    auto h = generate_csr(m, k, nnz);
    device_csr a(h);
    spblas::scale(2.5f, a.view);
    auto v = a.values.download();
    bool same = true;
    for (int i = 0; i < nnz; i++)
      same &= v[i] == h.values[i] * 2.5f;
    CHECK(same);
    std::vector<value_t> xh(k);
    for (int i = 0; i < k; i++)
      xh[i] = (value_t) (i % 7) - 3;
    dvec<value_t> x(xh);
    spblas::scale(-0.5f, std::span<value_t>(x.p, (size_t) k));
    auto xs = x.download();
    same = true;
    for (int i = 0; i < k; i++)
      same &= xs[i] == xh[i] * -0.5f;
    CHECK(same);
  }
}

// Value snapshot contract of the C ABI (spblas_gfx950.h): AUTO without OPT_VALUE_SNAPSHOT never chooses the plan
// that copies the values; with the option (what a matrix_opt operand sets) it may; a SLICED plan that is handed
// another value array than the one it copied refreshes its copy inside spblas_gfx950_spmv.  The reference reads the
// caller's values on every multiply (algorithms/multiply_impl.hpp:48-52).
static void test_value_snapshot_contract() {
  const int m = 260000, n = 1000000, per = 9;  // x = 4 MB, 2.3 M entries: a candidate for the sliced plan
  host_csr h;
  h.shape = spblas::index<index_t>(m, n);
  h.nnz = (offset_t) m * per;
  h.rowptr.resize(m + 1);
  for (int i = 0; i <= m; i++)
    h.rowptr[i] = (offset_t) i * per;
  std::mt19937 g(7);
  std::uniform_int_distribution<int> dc(0, n - 1);
  std::uniform_real_distribution<value_t> dv(0.5f, 1.5f);
  h.colind.resize(h.nnz);
  h.values.resize(h.nnz);
  for (offset_t p = 0; p < h.nnz; p++) {
    h.colind[p] = dc(g);
    h.values[p] = dv(g);
  }
  std::vector<value_t> xh(n);
  for (auto& v : xh)
    v = dv(g);
  device_csr a(h);
  dvec<value_t> x(xh), y(m);
  std::vector<value_t> v2(h.values);
  for (auto& v : v2)
    v *= 3.0f;
  dvec<value_t> values2(v2);
  auto row_ref = [&](int i, value_t s) {
    value_t r = 0;
    for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
      r += s * h.values[p] * xh[h.colind[p]];
    return r;
  };
  auto check_y = [&](value_t s) {
    auto yh = y.download();
    bool ok = true;
    for (int i = 0; i < m; i += 97)
      ok &= std::abs((double) yh[i] - (double) row_ref(i, s)) <= 2e-6 * std::abs((double) row_ref(i, s)) * per;
    CHECK(ok);
  };
  spblas_gfx950_handle_t hd = nullptr;
  CHECK(spblas_gfx950_create(&hd, nullptr) == SPBLAS_GFX950_STATUS_SUCCESS);
  const value_t one = 1, zero = 0;
  for (int mode = 0; mode < 3; ++mode) {  // 0: AUTO, no option; 1: AUTO + option; 2: explicit SLICED
    spblas_gfx950_plan_t plan = nullptr;
    CHECK(spblas_gfx950_set_option(hd, SPBLAS_GFX950_OPT_VALUE_SNAPSHOT, mode == 1) == SPBLAS_GFX950_STATUS_SUCCESS);
    CHECK(spblas_gfx950_spmv_plan_create(hd, &plan, m, n, h.nnz, a.rowptr.p, a.colind.p, a.values.p, SPBLAS_GFX950_I32,
                                         SPBLAS_GFX950_F32, mode == 2 ? SPBLAS_GFX950_SPMV_SLICED : SPBLAS_GFX950_SPMV_AUTO) ==
          SPBLAS_GFX950_STATUS_SUCCESS);
    std::int64_t info[12];
    CHECK(spblas_gfx950_plan_info(plan, info) == SPBLAS_GFX950_STATUS_SUCCESS);
    CHECK(info[0] == (mode == 0 ? SPBLAS_GFX950_SPMV_ROWBLOCK : SPBLAS_GFX950_SPMV_SLICED));
    CHECK(spblas_gfx950_spmv(hd, plan, SPBLAS_GFX950_OP_N, m, n, h.nnz, &one, a.rowptr.p, a.colind.p, a.values.p, x.p,
                             &zero, y.p, SPBLAS_GFX950_I32, SPBLAS_GFX950_F32) == SPBLAS_GFX950_STATUS_SUCCESS);
    check_y(1.0f);
    // same structure, another value array: the product must follow it (no update_values call)
    CHECK(spblas_gfx950_spmv(hd, plan, SPBLAS_GFX950_OP_N, m, n, h.nnz, &one, a.rowptr.p, a.colind.p, values2.p, x.p,
                             &zero, y.p, SPBLAS_GFX950_I32, SPBLAS_GFX950_F32) == SPBLAS_GFX950_STATUS_SUCCESS);
    check_y(3.0f);
    CHECK(spblas_gfx950_spmv(hd, plan, SPBLAS_GFX950_OP_N, m, n, h.nnz, &one, a.rowptr.p, a.colind.p, a.values.p, x.p,
                             &zero, y.p, SPBLAS_GFX950_I32, SPBLAS_GFX950_F32) == SPBLAS_GFX950_STATUS_SUCCESS);
    check_y(1.0f);
    CHECK(spblas_gfx950_plan_destroy(hd, plan) == SPBLAS_GFX950_STATUS_SUCCESS);
  }
  CHECK(spblas_gfx950_set_option(hd, SPBLAS_GFX950_OPT_VALUE_SNAPSHOT, 0) == SPBLAS_GFX950_STATUS_SUCCESS);
  (void) hipDeviceSynchronize();
  CHECK(spblas_gfx950_destroy(hd) == SPBLAS_GFX950_STATUS_SUCCESS);
  // the C++ layer: a plain view inspects to a structure-only plan, a matrix_opt may get the copying one; both
  // follow a value array rebound with csr_view::update (views/csr_view.hpp:36-49)
  std::span<value_t> xs(x.p, (size_t) n), ys(y.p, (size_t) m);
  for (int opt = 0; opt < 2; ++opt) {
    spblas::csr_view<value_t, index_t, offset_t> view(a.values.p, a.rowptr.p, a.colind.p, h.shape, h.nnz);
    spblas::matrix_opt view_opt(view);
    auto info = opt ? spblas::multiply_inspect(view_opt, xs, ys) : spblas::multiply_inspect(view, xs, ys);
    spblas::multiply(info, view, xs, ys);
    check_y(1.0f);
    view.update(std::span<value_t>(values2.p, (size_t) h.nnz), std::span<offset_t>(a.rowptr.p, (size_t) m + 1),
                std::span<index_t>(a.colind.p, (size_t) h.nnz));
    spblas::multiply(info, view, xs, ys);
    check_y(3.0f);
  }
}

// The execute call of a plan inside a HIP graph, straight through the C ABI (include/spblas_gfx950.h: "Graph capture"):
// an iterative solver records multiply(info, A, x, y) once and replays it; inspect runs outside the capture.
static void test_graph_capture() {
#define HIP_REQUIRE(expr)                                                  \
  do {                                                                     \
    const hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) {                                                \
      std::printf("FAIL %s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      ++g_fail;                                                            \
      return;                                                              \
    }                                                                      \
  } while (0)
  const int m = 30000, n = 50000;
  auto h = generate_csr(m, n, 400000, 17);
  device_csr a(h);
  hipStream_t stream = nullptr;
  HIP_REQUIRE(hipStreamCreate(&stream));
  spblas_gfx950_handle_t handle = nullptr;
  CHECK(spblas_gfx950_create(&handle, stream) == SPBLAS_GFX950_STATUS_SUCCESS);
  for (int alg : {SPBLAS_GFX950_SPMV_SLICED, SPBLAS_GFX950_SPMV_ROWBLOCK}) {
    spblas_gfx950_plan_t plan = nullptr;
    CHECK(spblas_gfx950_spmv_plan_create(handle, &plan, m, n, h.nnz, a.rowptr.p, a.colind.p, a.values.p,
                                         SPBLAS_GFX950_I32, SPBLAS_GFX950_F32, alg) == SPBLAS_GFX950_STATUS_SUCCESS);
    dvec<value_t> d_x(std::vector<value_t>(n, 0)), d_y(std::vector<value_t>(m, -1));
    const value_t alpha = 1, beta = 0;
    auto call = [&]() {
      return spblas_gfx950_spmv(handle, plan, SPBLAS_GFX950_OP_N, m, n, h.nnz, &alpha, a.rowptr.p, a.colind.p, a.values.p,
                                d_x.p, &beta, d_y.p, SPBLAS_GFX950_I32, SPBLAS_GFX950_F32);
    };
    CHECK(call() == SPBLAS_GFX950_STATUS_SUCCESS);  // one ordinary call first
    HIP_REQUIRE(hipStreamSynchronize(stream));
    // creating a plan on a capturing stream is refused, not recorded
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_REQUIRE(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    spblas_gfx950_plan_t refused = nullptr;
    const int rc_plan = spblas_gfx950_spmv_plan_create(handle, &refused, m, n, h.nnz, a.rowptr.p, a.colind.p, a.values.p,
                                                       SPBLAS_GFX950_I32, SPBLAS_GFX950_F32, alg);
    const int rc_call = call();
    HIP_REQUIRE(hipStreamEndCapture(stream, &graph));
    CHECK(rc_plan != SPBLAS_GFX950_STATUS_SUCCESS);
    CHECK(rc_call == SPBLAS_GFX950_STATUS_SUCCESS);
    if (refused)
      spblas_gfx950_plan_destroy(handle, refused);
    HIP_REQUIRE(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      std::vector<value_t> x(n);
      std::mt19937 g(100 + rep);
      std::uniform_int_distribution<int> d(0, 7);
      for (auto& v : x)
        v = (value_t) d(g);  // small integers: sums are exact
      HIP_REQUIRE(hipMemcpyAsync(d_x.p, x.data(), n * sizeof(value_t), hipMemcpyHostToDevice, stream));
      HIP_REQUIRE(hipMemsetAsync(d_y.p, 0xFF, m * sizeof(value_t), stream));
      HIP_REQUIRE(hipGraphLaunch(exec, stream));
      HIP_REQUIRE(hipStreamSynchronize(stream));
      auto y = d_y.download();
      int bad = 0;
      for (int i = 0; i < m; i++) {
        value_t ref = 0;
        for (auto p = h.rowptr[i]; p < h.rowptr[i + 1]; p++)
          ref += h.values[p] * x[h.colind[p]];
        bad += near_ref(ref, y[i]) ? 0 : 1;
      }
      CHECK(bad == 0);
    }
    HIP_REQUIRE(hipGraphExecDestroy(exec));
    HIP_REQUIRE(hipGraphDestroy(graph));
    CHECK(spblas_gfx950_plan_destroy(handle, plan) == SPBLAS_GFX950_STATUS_SUCCESS);
  }
  CHECK(spblas_gfx950_destroy(handle) == SPBLAS_GFX950_STATUS_SUCCESS);
  HIP_REQUIRE(hipStreamDestroy(stream));
#undef HIP_REQUIRE
}

int main() {
  test_graph_capture();
  test_scale();
  test_value_snapshot_contract();
  test_spmv();
  test_spmm();
  test_spgemm();
  test_spgemm_4args();
  test_add();
  test_triangular_solve();
  test_transpose();
  std::printf("%s: %d checks, %d failures\n", g_fail ? "FAILED" : "PASSED", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
