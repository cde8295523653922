// test stub: vk::hasParam / vk::getParam over the ros stub's parameter table
#ifndef RMD_TEST_STUB_VIKIT_PARAMS
#define RMD_TEST_STUB_VIKIT_PARAMS
#include <ros/ros.h>
namespace vk {
inline bool hasParam(const std::string& name) { return ros::stub::world().params.count(name) != 0; }
template <typename T>
T getParam(const std::string& name, const T& default_value) {
  const std::map<std::string, std::string>& p = ros::stub::world().params;
  const std::map<std::string, std::string>::const_iterator it = p.find(name);
  if (it == p.end()) return default_value;
  std::istringstream s(it->second);
  T v;
  s >> v;
  return v;
}
template <typename T>
T getParam(const std::string& name) { return getParam<T>(name, T()); }
}  // namespace vk
#endif
