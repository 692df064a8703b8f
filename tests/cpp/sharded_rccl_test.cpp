// -m gpu (tests/test_gpu_cpp.py): the C++ row-sharded SpMV over RCCL (include/spblas/vendor/gfx950/sharded_spmv.hpp) with
// the ONE rank a one-GPU box allows: communicator of one rank, equal and uneven bounds paths (the uneven path with one rank
// degenerates to the local SpMV; the bounds arithmetic is checked on the host), plan from inspect, y against a host loop.
// With SHARDED_RANKS=N and the usual RANK / WORLD_SIZE variables it runs as one of N processes on N GPUs (id exchanged
// through a file), which is how a maintainer would try it on a multi-GPU node; the test box runs it with one.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include <spblas/vendor/gfx950/fused_sharded_spmv.hpp>
#include <spblas/vendor/gfx950/sharded_spmv.hpp>

#define HIP_OK(e)                                                                  \
  do {                                                                             \
    if ((e) != hipSuccess) {                                                       \
      std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__);            \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

int main() {
  using T = float;
  const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
  const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  HIP_OK(hipSetDevice(rank % ndev));
  ncclUniqueId id;
  const char* idfile = std::getenv("SHARDED_ID_FILE");
  if (rank == 0) {
    if (ncclGetUniqueId(&id) != ncclSuccess)
      return 3;
    if (idfile) {
      std::ofstream f(idfile, std::ios::binary);
      f.write(reinterpret_cast<const char*>(&id), sizeof(id));
    }
  } else {
    for (int tries = 0; tries < 600; ++tries) {
      std::ifstream f(idfile ? idfile : "", std::ios::binary);
      if (f && f.read(reinterpret_cast<char*>(&id), sizeof(id)))
        break;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  ncclComm_t comm;
  if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess)
    return 4;
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  // the global matrix, generated identically on every rank: m x n, ragged rows, a few long ones
  const std::int64_t m = 50000, n = 70000;
  std::vector<std::int32_t> rowptr(m + 1, 0), colind;
  std::vector<T> values, x(n);
  unsigned long long sd = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return (unsigned) (sd >> 33); };
  for (std::int64_t r = 0; r < m; ++r) {
    int len = rnd() % 12;
    if (r % 9973 == 0)
      len = 3000;
    for (int k = 0; k < len; ++k) {
      colind.push_back((std::int32_t) (rnd() % n));
      values.push_back((T) (rnd() % 1000) / 1000.f - 0.5f);
    }
    rowptr[r + 1] = (std::int32_t) colind.size();
  }
  for (auto& v : x)
    v = (T) (rnd() % 1000) / 1000.f - 0.5f;
  std::vector<double> want(m, 0.0), scale(m, 0.0);
  for (std::int64_t r = 0; r < m; ++r)
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const double t = (double) values[p] * x[colind[p]];
      want[r] += t;
      scale[r] += std::fabs(t);
    }
  int failed = 0;
  using op_t = spblas::__gfx950::sharded_spmv_t<T, std::int32_t>;
  const auto by_nnz = op_t::partition_rows_by_nnz(rowptr.data(), m, 4, 0);  // (row_weight 0: the entries alone)
  if (!(by_nnz.size() == 5 && by_nnz[0] == 0 && by_nnz[4] == m && by_nnz[1] <= by_nnz[2] && by_nnz[2] <= by_nnz[3])) {
    std::fprintf(stderr, "FAILED: partition_rows_by_nnz shape\n");
    ++failed;
  }
  for (int g = 1; g < 4; ++g) {  // each part holds about a quarter of the entries (a row of 3 000 is the granularity)
    const double share = (double) (rowptr[by_nnz[g]] - rowptr[by_nnz[g - 1]]) / rowptr[m];
    if (std::fabs(share - 0.25) > 0.02) {
      std::fprintf(stderr, "FAILED: partition_rows_by_nnz balance %f\n", share);
      ++failed;
    }
  }
  for (int uneven = 0; uneven < 2; ++uneven) {
    std::vector<std::int64_t> bounds = uneven ? op_t::partition_rows_by_nnz(rowptr.data(), m, world)
                                              : std::vector<std::int64_t>();
    if (!uneven) {
      bounds.resize(world + 1);
      for (int r = 0; r <= world; ++r)
        bounds[r] = m * r / world;
    }
    const std::int64_t r0 = bounds[rank], r1 = bounds[rank + 1], lnnz = rowptr[r1] - rowptr[r0];
    std::vector<std::int32_t> lrp(r1 - r0 + 1);
    for (std::int64_t r = r0; r <= r1; ++r)
      lrp[r - r0] = rowptr[r] - rowptr[r0];
    std::int32_t *d_rp, *d_ci;
    T *d_v, *d_x, *d_y;
    HIP_OK(hipMalloc(&d_rp, lrp.size() * 4));
    HIP_OK(hipMalloc(&d_ci, (lnnz + 1) * 4));
    HIP_OK(hipMalloc(&d_v, (lnnz + 1) * sizeof(T)));
    HIP_OK(hipMalloc(&d_x, n * sizeof(T)));
    HIP_OK(hipMalloc(&d_y, m * sizeof(T)));
    HIP_OK(hipMemcpy(d_rp, lrp.data(), lrp.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ci, colind.data() + rowptr[r0], lnnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_v, values.data() + rowptr[r0], lnnz * sizeof(T), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_x, x.data(), n * sizeof(T), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_y, 0xFF, m * sizeof(T)));
    op_t op(comm, rank, world, bounds, stream);
    op.inspect(n, lnnz, d_rp, d_ci, d_v, true);
    for (int rep = 0; rep < 2; ++rep)
      op.multiply(T(2), n, lnnz, d_rp, d_ci, d_v, d_x, d_y);
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<T> y(m);
    HIP_OK(hipMemcpy(y.data(), d_y, m * sizeof(T), hipMemcpyDeviceToHost));
    int bad = 0;
    for (std::int64_t r = 0; r < m; ++r)
      if (!(std::fabs((double) y[r] - 2.0 * want[r]) <= 2e-6 * 2.0 * scale[r] + 1e-30))
        ++bad;
    if (bad) {
      std::fprintf(stderr, "FAILED: rank %d, %s shards: %d rows of the gathered y differ\n", rank, uneven ? "uneven" : "equal", bad);
      ++failed;
    }
    (void) hipFree(d_rp); (void) hipFree(d_ci); (void) hipFree(d_v); (void) hipFree(d_x); (void) hipFree(d_y);
  }
  // The fused exchange from C++ (fused_sharded_spmv.hpp: peer stores from the reduce kernels into hipIpc-mapped copies of y,
  // device-side step barrier; RCCL only for the handle exchange): equal shards of a square matrix large enough for the
  // SLICED plan, four steps with different vectors (both copies of y written twice), every step compared BIT FOR BIT with
  // the RCCL all-gather operator running the same kind of plan on the same shard.
  {
    const std::int64_t per = 60000, mf = per * world, nf = mf;
    std::vector<std::int32_t> frp(per + 1, 0), fci;
    std::vector<T> fv, fx(nf);
    sd = 0x1234567ull + 77ull * (unsigned long long) rank;
    for (std::int64_t r = 0; r < per; ++r) {
      const int len = 6 + (int) (rnd() % 9);
      for (int k = 0; k < len; ++k) {
        fci.push_back((std::int32_t) (rnd() % nf));
        fv.push_back((T) (rnd() % 1000) / 1000.f - 0.5f);
      }
      frp[r + 1] = (std::int32_t) fci.size();
    }
    sd = 42;
    for (auto& v : fx)
      v = (T) (rnd() % 1000) / 1000.f - 0.5f;
    const std::int64_t fnnz = (std::int64_t) fci.size();
    std::int32_t *d_rp, *d_ci;
    T *d_v, *d_x, *d_y;
    HIP_OK(hipMalloc(&d_rp, frp.size() * 4));
    HIP_OK(hipMalloc(&d_ci, fnnz * 4));
    HIP_OK(hipMalloc(&d_v, fnnz * sizeof(T)));
    HIP_OK(hipMalloc(&d_x, nf * sizeof(T)));
    HIP_OK(hipMalloc(&d_y, mf * sizeof(T)));
    HIP_OK(hipMemcpy(d_rp, frp.data(), frp.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ci, fci.data(), fnnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_v, fv.data(), fnnz * sizeof(T), hipMemcpyHostToDevice));
    std::vector<std::int64_t> bounds(world + 1);
    for (int r = 0; r <= world; ++r)
      bounds[r] = per * r;
    try {
      spblas::__gfx950::fused_sharded_spmv_t<T, std::int32_t> fop(comm, rank, world, per, stream, 5000);
      fop.inspect(nf, fnnz, d_rp, d_ci, d_v);
      op_t rop(comm, rank, world, bounds, stream);
      rop.inspect(nf, fnnz, d_rp, d_ci, d_v, true);
      std::vector<T> yf(mf), yr(mf);
      for (int stepno = 0; stepno < 4; ++stepno) {
        for (std::int64_t i = 0; i < nf; ++i)
          fx[i] = fx[i] * (T) (stepno % 2 ? -0.5 : 1.25) + (T) 0.125 * (T) stepno;
        HIP_OK(hipMemcpy(d_x, fx.data(), nf * sizeof(T), hipMemcpyHostToDevice));
        const T* y_fused = fop.step(T(2), d_x);
        HIP_OK(hipStreamSynchronize(stream));
        fop.check_status();
        HIP_OK(hipMemcpy(yf.data(), y_fused, mf * sizeof(T), hipMemcpyDeviceToHost));
        HIP_OK(hipMemset(d_y, 0xFF, mf * sizeof(T)));
        rop.multiply(T(2), nf, fnnz, d_rp, d_ci, d_v, d_x, d_y);
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(yr.data(), d_y, mf * sizeof(T), hipMemcpyDeviceToHost));
        std::int64_t bad = 0, close_enough = 0;
        for (std::int64_t r = 0; r < mf; ++r) {
          bad += std::memcmp(&yf[r], &yr[r], sizeof(T)) != 0;
          close_enough += std::fabs((double) yf[r] - (double) yr[r]) <= 1e-4 * std::fabs((double) yr[r]) + 2e-5;
        }
        // (the two operators may run different plans -- tiles here, nnz windows there -- and sum a row in different orders:
        // the comparison is to rounding, |y| = O(1); a wiring error -- wrong offset, missing rows, stale copy -- is O(1))
        if (close_enough != mf) {
          std::fprintf(stderr, "FAILED: rank %d, fused step %d: %lld rows differ from the RCCL operator beyond rounding (%lld not bit-equal)\n",
                       rank, stepno, (long long) (mf - close_enough), (long long) bad);
          ++failed;
        }
      }
    } catch (const std::exception& e) {
      std::fprintf(stderr, "FAILED: rank %d, fused operator: %s\n", rank, e.what());
      ++failed;
    }
    (void) hipFree(d_rp); (void) hipFree(d_ci); (void) hipFree(d_v); (void) hipFree(d_x); (void) hipFree(d_y);
  }
  ncclCommDestroy(comm);
  if (failed == 0 && rank == 0)
    std::printf("SHARDED_RCCL_OK ranks=%d\n", world);
  return failed ? 1 : 0;
}
