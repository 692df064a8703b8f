// main() for the reference's device tests compiled against the stand-in <gtest/gtest.h> (stubs/gtest/gtest.h):
// runs every registered TEST, prints one line per test in GoogleTest's style and returns the number of failed tests.
#include <gtest/gtest.h>

int main() {
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
    if (testing_stub::failures_in_current_test() == 0) {
      std::printf("[       OK ] %s.%s\n", t.suite, t.name);
    } else {
      std::printf("[  FAILED  ] %s.%s (%d expectations)\n", t.suite, t.name, testing_stub::failures_in_current_test());
      ++failed;
    }
  }
  std::printf("[==========] %zu tests ran, %d failed.\n", tests.size(), failed);
  return failed;
}
