// TEST-RUN INFRASTRUCTURE ONLY (tests/compile_check/build_dropin.py: build_reference_device_tests()).
// The image has no GoogleTest.  This is the handful of macros the reference's device tests use -- TEST,
// EXPECT_EQ, EXPECT_NE, EXPECT_LE, EXPECT_NEAR -- with a registry and a main() (gtest_main.cpp), so that those test files can be compiled
// UNMODIFIED from the reference tree against the gfx950 backend.  Not GoogleTest, pins nothing.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing_stub {
struct test_case {
  const char* suite;
  const char* name;
  void (*fn)();
};
inline std::vector<test_case>& registry() {
  static std::vector<test_case> r;
  return r;
}
inline int& failures_in_current_test() {
  static int f = 0;
  return f;
}
inline int& reported() {
  static int f = 0;
  return f;
}
struct registrar {
  registrar(const char* suite, const char* name, void (*fn)()) { registry().push_back({suite, name, fn}); }
};
template <typename A, typename B>
void report(const char* kind, const char* ea, const char* eb, const A& a, const B& b, const char* file, int line) {
  ++failures_in_current_test();
  if (reported()++ < 20)  // the first few in full, the rest counted
    std::cerr << file << ":" << line << ": " << kind << "(" << ea << ", " << eb << ") failed: " << a << " vs " << b << "\n";
}
} // namespace testing_stub

#define TEST(suite, name)                                                                             \
  static void suite##_##name##_body();                                                                \
  static ::testing_stub::registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_body); \
  static void suite##_##name##_body()

#define EXPECT_EQ(a, b)                                                                       \
  do {                                                                                        \
    auto&& va_ = (a);                                                                         \
    auto&& vb_ = (b);                                                                         \
    if (!(va_ == vb_))                                                                        \
      ::testing_stub::report("EXPECT_EQ", #a, #b, va_, vb_, __FILE__, __LINE__);              \
  } while (0)

#define EXPECT_NE(a, b)                                                                       \
  do {                                                                                        \
    auto&& va_ = (a);                                                                         \
    auto&& vb_ = (b);                                                                         \
    if (!(va_ != vb_))                                                                        \
      ::testing_stub::report("EXPECT_NE", #a, #b, va_, vb_, __FILE__, __LINE__);              \
  } while (0)

#define EXPECT_LE(a, b)                                                                       \
  do {                                                                                        \
    auto&& va_ = (a);                                                                         \
    auto&& vb_ = (b);                                                                         \
    if (!(va_ <= vb_))                                                                        \
      ::testing_stub::report("EXPECT_LE", #a, #b, va_, vb_, __FILE__, __LINE__);              \
  } while (0)

#define EXPECT_NEAR(a, b, tol)                                                                \
  do {                                                                                        \
    auto&& va_ = (a);                                                                         \
    auto&& vb_ = (b);                                                                         \
    if (!(std::abs(va_ - vb_) <= (tol)))                                                      \
      ::testing_stub::report("EXPECT_NEAR", #a, #b, va_, vb_, __FILE__, __LINE__);            \
  } while (0)
