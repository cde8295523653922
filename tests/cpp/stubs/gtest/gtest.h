// test stub: the part of googletest the reference's test/*.cpp use -- TEST, ASSERT_/EXPECT_ {EQ, NEAR, FLOAT_EQ}, FAIL, message
// streaming, InitGoogleTest, RUN_ALL_TESTS -- so that the reference's own test sources compile and run unmodified.
// ASSERT_FLOAT_EQ is googletest's 4-ULP comparison on the values converted to float.
#ifndef RMD_TEST_STUB_GTEST
#define RMD_TEST_STUB_GTEST
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing {
namespace stub {
struct TestInfo {
  const char* suite;
  const char* name;
  void (*fn)();
};
inline std::vector<TestInfo>& registry() {
  static std::vector<TestInfo> r;
  return r;
}
struct State {
  int failures_in_test = 0;
};
inline State& state() {
  static State s;
  return s;
}
struct Registrar {
  Registrar(const char* suite, const char* name, void (*fn)()) {
    const TestInfo t = {suite, name, fn};
    registry().push_back(t);
  }
};
class Message {
 public:
  template <typename T> Message& operator<<(const T& v) { s_ << v; return *this; }
  Message& operator<<(std::ostream& (*manip)(std::ostream&)) { s_ << manip; return *this; }
  std::string str() const { return s_.str(); }
 private:
  std::ostringstream s_;
};
class Helper {
 public:
  Helper(const char* file, int line, const std::string& text) : file_(file), line_(line), text_(text) {}
  void operator=(const Message& m) const {
    ++state().failures_in_test;
    std::cout << file_ << ":" << line_ << ": Failure\n" << text_ << "\n" << m.str() << std::endl;
  }
 private:
  const char* file_;
  int line_;
  std::string text_;
};
template <typename A, typename B>
bool eq(const A& a, const B& b) { return a == b; }
inline bool float_eq(float a, float b) {  // googletest FloatingPoint<float>::AlmostEquals: at most 4 ULPs apart, NaN never equal
  if (a != a || b != b) return false;
  uint32_t ia, ib;
  memcpy(&ia, &a, 4);
  memcpy(&ib, &b, 4);
  const uint32_t ba = (ia & 0x80000000u) ? ~ia + 1 : ia | 0x80000000u, bb = (ib & 0x80000000u) ? ~ib + 1 : ib | 0x80000000u;
  return (ba >= bb ? ba - bb : bb - ba) <= 4u;
}
template <typename A, typename B>
std::string describe(const char* what, const char* ea, const char* eb, const A& a, const B& b) {
  std::ostringstream s;
  s.precision(9);
  s << what << "(" << ea << ", " << eb << "): " << a << " vs " << b;
  return s.str();
}
}  // namespace stub

inline void InitGoogleTest(int*, char**) {}
}  // namespace testing

#define TEST(suite, name)                                                                                        \
  static void suite##_##name##_Test();                                                                           \
  static ::testing::stub::Registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_Test);          \
  static void suite##_##name##_Test()

#define GSTUB_FAILURE_(text) ::testing::stub::Helper(__FILE__, __LINE__, text) = ::testing::stub::Message()
#define GSTUB_ASSERT_(ok, text) if (ok) ; else return GSTUB_FAILURE_(text)
#define GSTUB_EXPECT_(ok, text) if (ok) ; else GSTUB_FAILURE_(text)

#define FAIL() return GSTUB_FAILURE_("Failed")
#define ASSERT_EQ(a, b) GSTUB_ASSERT_(::testing::stub::eq((a), (b)), ::testing::stub::describe("ASSERT_EQ", #a, #b, (a), (b)))
#define EXPECT_EQ(a, b) GSTUB_EXPECT_(::testing::stub::eq((a), (b)), ::testing::stub::describe("EXPECT_EQ", #a, #b, (a), (b)))
#define ASSERT_TRUE(c) GSTUB_ASSERT_((c), std::string("ASSERT_TRUE(") + #c + ")")
#define EXPECT_TRUE(c) GSTUB_EXPECT_((c), std::string("EXPECT_TRUE(") + #c + ")")
#define ASSERT_FLOAT_EQ(a, b) \
  GSTUB_ASSERT_(::testing::stub::float_eq(static_cast<float>(a), static_cast<float>(b)), ::testing::stub::describe("ASSERT_FLOAT_EQ", #a, #b, (a), (b)))
#define EXPECT_FLOAT_EQ(a, b) \
  GSTUB_EXPECT_(::testing::stub::float_eq(static_cast<float>(a), static_cast<float>(b)), ::testing::stub::describe("EXPECT_FLOAT_EQ", #a, #b, (a), (b)))
#define ASSERT_NEAR(a, b, tol) \
  GSTUB_ASSERT_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol), ::testing::stub::describe("ASSERT_NEAR", #a, #b, (a), (b)))
#define EXPECT_NEAR(a, b, tol) \
  GSTUB_EXPECT_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol), ::testing::stub::describe("EXPECT_NEAR", #a, #b, (a), (b)))

inline int RUN_ALL_TESTS() {
  int failed = 0;
  const std::vector< ::testing::stub::TestInfo>& tests = ::testing::stub::registry();
  std::cout << "[==========] Running " << tests.size() << " tests." << std::endl;
  for (size_t i = 0; i < tests.size(); ++i) {
    std::cout << "[ RUN      ] " << tests[i].suite << "." << tests[i].name << std::endl;
    ::testing::stub::state().failures_in_test = 0;
    tests[i].fn();
    const int f = ::testing::stub::state().failures_in_test;
    std::cout << (f ? "[  FAILED  ] " : "[       OK ] ") << tests[i].suite << "." << tests[i].name;
    if (f) std::cout << " (" << f << " failed checks)";
    std::cout << std::endl;
    failed += f != 0;
  }
  std::cout << "[==========] " << tests.size() << " tests ran.\n[  PASSED  ] " << tests.size() - failed << " tests." << std::endl;
  if (failed) std::cout << "[  FAILED  ] " << failed << " tests." << std::endl;
  return failed ? 1 : 0;
}
#endif
