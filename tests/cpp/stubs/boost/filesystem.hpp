// test stub: boost::filesystem::path as test/dataset.cpp uses it (construct, operator/, string())
#ifndef RMD_TEST_STUB_BOOST_FILESYSTEM
#define RMD_TEST_STUB_BOOST_FILESYSTEM
#include <iostream>
#include <sstream>
#include <string>
namespace boost {
namespace filesystem {
class path {
 public:
  path() {}
  path(const std::string& s) : s_(s) {}
  path(const char* s) : s_(s) {}
  const std::string& string() const { return s_; }
  path operator/(const path& rhs) const {
    if (s_.empty()) return rhs;
    return path(s_.back() == '/' ? s_ + rhs.s_ : s_ + "/" + rhs.s_);
  }
 private:
  std::string s_;
};
inline path operator/(const path& lhs, const std::string& rhs) { return lhs / path(rhs); }
inline path operator/(const path& lhs, const char* rhs) { return lhs / path(rhs); }
}  // namespace filesystem
}  // namespace boost
#endif
