// test stub: image_transport::ImageTransport::advertise / Publisher::publish onto the file "topics" of the ros stub
#ifndef RMD_TEST_STUB_IMAGE_TRANSPORT
#define RMD_TEST_STUB_IMAGE_TRANSPORT
#include <ros/ros.h>
#include <sensor_msgs/image_encodings.h>
namespace image_transport {
class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string& topic) : pub_(topic) {}
  void publish(const sensor_msgs::ImagePtr& msg) const { pub_.publish(*msg); }
 private:
  ros::Publisher pub_;
};
class ImageTransport {
 public:
  explicit ImageTransport(const ros::NodeHandle&) {}
  Publisher advertise(const std::string& topic, uint32_t) { return Publisher(topic); }
};
}  // namespace image_transport
#endif
