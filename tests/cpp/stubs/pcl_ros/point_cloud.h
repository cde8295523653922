// test stub: pcl::PointXYZI, pcl::PointCloud<> and pcl_conversions::toPCL as src/publisher.cpp uses them
#ifndef RMD_TEST_STUB_PCL_ROS
#define RMD_TEST_STUB_PCL_ROS
#include <ros/ros.h>
#define PCL_MAJOR_VERSION 1
#define PCL_MINOR_VERSION 8
namespace pcl {
struct PointXYZI {
  float x, y, z, intensity;
};
struct PCLHeader {
  std::string frame_id;
  uint64_t stamp;
  PCLHeader() : stamp(0) {}
};
template <class PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  PCLHeader header;
  std::vector<PointT> points;
  void push_back(const PointT& p) { points.push_back(p); }
  bool empty() const { return points.empty(); }
  size_t size() const { return points.size(); }
  void stubWrite(FILE* f) const {  // int32 n, then n x (x, y, z, intensity) float32
    const int n = static_cast<int>(points.size());
    fwrite(&n, sizeof(int), 1, f);
    fwrite(points.data(), sizeof(PointT), points.size(), f);
  }
};
}  // namespace pcl
namespace pcl_conversions {
inline void toPCL(const ros::Time& t, uint64_t& stamp) { stamp = static_cast<uint64_t>(t.sec * 1e6); }
}  // namespace pcl_conversions
#endif
