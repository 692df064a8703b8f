// main() for the reference's HOST tests (test/gtest/{spmv,spmm,spgemm,add,transpose,triangular_solve}_test.cpp ...)
// run against the gfx950 DEVICE backend.  Those tests keep their matrices in std::vector and read the results straight
// from host memory; a device backend needs device-visible arrays.  Nothing in the tests or in the backend is changed
// for that: this file replaces the global operator new / delete of the TEST BINARY with a bump allocator over one
// slab of pinned, device-mapped host memory (hipHostMalloc), so every std::vector of the tests is memory the GPU can
// read and write over PCIe, and tests/test_gpu_dropin.py runs the binary with AMD_SERIALIZE_KERNEL=3 /
// AMD_SERIALIZE_COPY=3 so that every launch has completed when the call returns (the tests never synchronise: they
// were written for CPU backends).  Test infrastructure only; slow by construction (zero-copy) and irrelevant to the
// product's performance.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>

#include <hip/hip_runtime_api.h>

#include <gtest/gtest.h>

namespace {
char* g_slab = nullptr;
std::size_t g_cap = 0;
std::atomic<std::size_t> g_off{0};

void* slab_alloc(std::size_t n) {
  if (g_slab) {
    const std::size_t need = (n + 127) & ~static_cast<std::size_t>(127);
    const std::size_t at = g_off.fetch_add(need);
    if (at + need <= g_cap)
      return g_slab + at;
  }
  void* p = std::malloc(n ? n : 1);  // before the slab exists (static initialisation) or after it is exhausted
  if (!p)
    throw std::bad_alloc();
  return p;
}
void slab_free(void* p) noexcept {
  if (!p)
    return;
  if (g_slab && static_cast<char*>(p) >= g_slab && static_cast<char*>(p) < g_slab + g_cap)
    return;  // bump allocator: the slab is released as a whole at exit
  std::free(p);
}
} // namespace

void* operator new(std::size_t n) { return slab_alloc(n); }
void* operator new[](std::size_t n) { return slab_alloc(n); }
void* operator new(std::size_t n, std::align_val_t) { return slab_alloc(n); }
void* operator new[](std::size_t n, std::align_val_t) { return slab_alloc(n); }
void operator delete(void* p) noexcept { slab_free(p); }
void operator delete[](void* p) noexcept { slab_free(p); }
void operator delete(void* p, std::size_t) noexcept { slab_free(p); }
void operator delete[](void* p, std::size_t) noexcept { slab_free(p); }
void operator delete(void* p, std::align_val_t) noexcept { slab_free(p); }
void operator delete[](void* p, std::align_val_t) noexcept { slab_free(p); }
void operator delete(void* p, std::size_t, std::align_val_t) noexcept { slab_free(p); }
void operator delete[](void* p, std::size_t, std::align_val_t) noexcept { slab_free(p); }

int main() {
  const char* mb = std::getenv("DROPIN_PINNED_MB");
  const std::size_t cap = static_cast<std::size_t>(mb && *mb ? std::atoll(mb) : 3072) << 20;
  void* slab = nullptr;
  if (hipHostMalloc(&slab, cap, hipHostMallocDefault) != hipSuccess || !slab) {
    std::fprintf(stderr, "hipHostMalloc of the %zu MiB test slab failed\n", cap >> 20);
    return 3;
  }
  g_cap = cap;
  g_slab = static_cast<char*>(slab);  // from here on every heap allocation of the process is device-visible
  int failed = 0;
  const auto& tests = testing_stub::registry();
  std::printf("[==========] Running %zu tests.\n", tests.size());
  for (const auto& t : tests) {
    testing_stub::failures_in_current_test() = 0;
    std::printf("[ RUN      ] %s.%s\n", t.suite, t.name);
    std::fflush(stdout);
    try {
      t.fn();
    } catch (const std::exception& e) {
      std::fprintf(stderr, "exception: %s\n", e.what());
      ++testing_stub::failures_in_current_test();
    }
    (void) hipDeviceSynchronize();
    if (testing_stub::failures_in_current_test() == 0) {
      std::printf("[       OK ] %s.%s\n", t.suite, t.name);
    } else {
      std::printf("[  FAILED  ] %s.%s (%d expectations)\n", t.suite, t.name, testing_stub::failures_in_current_test());
      ++failed;
    }
  }
  std::printf("[==========] %zu tests ran, %d failed. (%zu MiB of the pinned slab used)\n", tests.size(), failed,
              g_off.load() >> 20);
  std::fflush(stdout);
  std::_Exit(failed);  // skip static destructors: the heap they would walk is the slab
}
