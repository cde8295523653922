// test stub: cv_bridge::CvImage / toCvShare over the cv::Mat stand-in
#ifndef RMD_TEST_STUB_CV_BRIDGE
#define RMD_TEST_STUB_CV_BRIDGE
#include <cstring>
#include <stdexcept>
#include <opencv2/opencv.hpp>
#include <sensor_msgs/image_encodings.h>
namespace cv_bridge {
class Exception : public std::runtime_error {
 public:
  explicit Exception(const std::string& what) : std::runtime_error(what) {}
};
class CvImage {
 public:
  std_msgs::Header header;
  std::string encoding;
  cv::Mat image;
  sensor_msgs::ImagePtr toImageMsg() const {
    sensor_msgs::ImagePtr msg(new sensor_msgs::Image);
    msg->header = header;
    msg->encoding = encoding;
    msg->height = image.rows; msg->width = image.cols; msg->elem_bytes = cv::stub_elem_size(image.type());
    msg->data.assign(image.data, image.data + static_cast<size_t>(image.rows) * image.cols * msg->elem_bytes);
    return msg;
  }
};
typedef std::shared_ptr<CvImage> CvImagePtr;
typedef std::shared_ptr<CvImage const> CvImageConstPtr;
template <class Tracked>
CvImageConstPtr toCvShare(const sensor_msgs::Image& source, const Tracked& /*tracked_object*/, const std::string& encoding) {
  if (encoding != source.encoding) throw Exception("stub cv_bridge: no conversion from " + source.encoding + " to " + encoding);
  CvImagePtr out(new CvImage);
  out->header = source.header;
  out->encoding = encoding;
  out->image.create(source.height, source.width, CV_8UC1);
  memcpy(out->image.data, source.data.data(), source.data.size());
  return out;
}
}  // namespace cv_bridge
#endif
