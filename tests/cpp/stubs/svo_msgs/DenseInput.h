// test stub: svo_msgs/DenseInput (header, frame_id, pose, image, min_depth, max_depth), filled from the ros stub's bag
#ifndef RMD_TEST_STUB_SVO_MSGS
#define RMD_TEST_STUB_SVO_MSGS
#include <ros/ros.h>
#include <sensor_msgs/image_encodings.h>
namespace geometry_msgs {
struct Point { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Pose {
  Point position;
  Quaternion orientation;
};
}  // namespace geometry_msgs
namespace svo_msgs {
struct DenseInput {
  std_msgs::Header header;
  uint32_t frame_id;
  geometry_msgs::Pose pose;
  sensor_msgs::Image image;
  float min_depth, max_depth;
  void stubFill(const ros::stub::BagMessage& b) {
    frame_id = 0;
    pose.position.x = b.position[0]; pose.position.y = b.position[1]; pose.position.z = b.position[2];
    pose.orientation.w = b.orientation_wxyz[0]; pose.orientation.x = b.orientation_wxyz[1];
    pose.orientation.y = b.orientation_wxyz[2]; pose.orientation.z = b.orientation_wxyz[3];
    image.height = b.height; image.width = b.width; image.elem_bytes = 1;
    image.encoding = sensor_msgs::image_encodings::MONO8;
    image.data = b.image;
    min_depth = b.min_depth; max_depth = b.max_depth;
  }
};
typedef std::shared_ptr<DenseInput> DenseInputPtr;
typedef std::shared_ptr<DenseInput const> DenseInputConstPtr;
}  // namespace svo_msgs
#endif
