// test stub: sensor_msgs::Image and the three encodings the reference names
#ifndef RMD_TEST_STUB_SENSOR_MSGS
#define RMD_TEST_STUB_SENSOR_MSGS
#include <ros/ros.h>
namespace sensor_msgs {
namespace image_encodings {
const std::string MONO8 = "mono8";
const std::string BGR8 = "bgr8";
const std::string TYPE_32FC1 = "32FC1";
}  // namespace image_encodings
struct Image {
  std_msgs::Header header;
  int height, width, elem_bytes;
  std::string encoding;
  std::vector<unsigned char> data;
  Image() : height(0), width(0), elem_bytes(1) {}
  void stubWrite(FILE* f) const {  // int32 rows, cols, bytes per pixel; then the pixels
    const int hdr[3] = {height, width, elem_bytes};
    fwrite(hdr, sizeof(int), 3, f);
    fwrite(data.data(), 1, data.size(), f);
  }
};
typedef std::shared_ptr<Image> ImagePtr;
typedef std::shared_ptr<Image const> ImageConstPtr;
}  // namespace sensor_msgs
#endif
