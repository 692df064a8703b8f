// Link-in for the reference's HOST EXAMPLES (examples/*.cpp: std::vector operands, their own main()) when they are
// built against the gfx950 device backend: the same pinned, device-visible heap as gtest_main_pinned.cpp, set up by a
// constructor that runs before main().  Test infrastructure only.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <new>

#include <hip/hip_runtime_api.h>

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
  void* p = std::malloc(n ? n : 1);
  if (!p)
    throw std::bad_alloc();
  return p;
}
void slab_free(void* p) noexcept {
  if (!p)
    return;
  if (g_slab && static_cast<char*>(p) >= g_slab && static_cast<char*>(p) < g_slab + g_cap)
    return;
  std::free(p);
}

__attribute__((constructor(65000))) void pinned_heap_init() {
  const char* mb = std::getenv("DROPIN_PINNED_MB");
  const std::size_t cap = static_cast<std::size_t>(mb && *mb ? std::atoll(mb) : 1024) << 20;
  void* slab = nullptr;
  if (hipHostMalloc(&slab, cap, hipHostMallocDefault) != hipSuccess || !slab) {
    std::fprintf(stderr, "pinned_heap: hipHostMalloc of %zu MiB failed\n", cap >> 20);
    std::_Exit(3);
  }
  g_cap = cap;
  g_slab = static_cast<char*>(slab);
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
