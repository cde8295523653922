// test stub: the slice of OpenCV the reference's host sources touch (src/depthmap.cpp, test/dataset.cpp, test/dataset_main.cpp).
// cv::Mat is a reference-counted row-major buffer; convertTo / remap / initUndistortRectifyMap follow OpenCV's published
// arithmetic the way the library restates it (the map computation IS the library's host function); imshow writes the image to
// $RMD_STUB_IMSHOW_DIR/<window>.bin so that a test can look at what the program displayed; waitKey returns at once.
#ifndef RMD_TEST_STUB_OPENCV
#define RMD_TEST_STUB_OPENCV
#include <chrono>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include <rmd_hip.h>

#define CV_8U 0
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16SC2 CV_MAKETYPE(CV_16S, 2)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_INTER_LINEAR 1
#define CV_LOAD_IMAGE_GRAYSCALE 0
#define CV_GRAY2BGR 8

namespace cv {

struct Vec3b {
  unsigned char val[3];
  unsigned char& operator[](int i) { return val[i]; }
  const unsigned char& operator[](int i) const { return val[i]; }
};

struct Scalar {
  double val[4];
  Scalar(double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0) : val{v0, v1, v2, v3} {}
};
struct Point {
  int x, y;
  Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {}
};
enum { EVENT_LBUTTONDOWN = 1 };
typedef void (*MouseCallback)(int event, int x, int y, int flags, void* userdata);
inline void setMouseCallback(const std::string&, MouseCallback, void* = NULL) {}

// cv::RNG: OpenCV's multiply-with-carry generator (state 2^32 - 1 by default, multiplier 4164903690)
class RNG {
 public:
  uint64_t state;
  RNG() : state(0xffffffffu) {}
  unsigned next() {
    state = static_cast<uint64_t>(static_cast<unsigned>(state)) * 4164903690u + static_cast<unsigned>(state >> 32);
    return static_cast<unsigned>(state);
  }
  int uniform(int a, int b) { return a == b ? a : static_cast<int>(next() % static_cast<unsigned>(b - a) + a); }
  float uniform(float a, float b) { return static_cast<float>(next()) * 2.3283064365386963e-10f * (b - a) + a; }
};

struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};

inline int stub_elem_size(int type) {
  static const int depth_bytes[7] = {1, 1, 2, 2, 4, 4, 8};
  return depth_bytes[type & 7] * ((type >> 3) + 1);
}
template <typename T> struct StubType;
template <> struct StubType<float> { static const int value = CV_32FC1; };
template <> struct StubType<int> { static const int value = CV_32SC1; };
template <> struct StubType<double> { static const int value = CV_64FC1; };
template <> struct StubType<unsigned char> { static const int value = CV_8UC1; };

class Mat {
 public:
  int rows, cols;
  unsigned char* data;
  Mat() : rows(0), cols(0), data(NULL), type_(CV_8UC1) {}
  Mat(int r, int c, int type) : rows(0), cols(0), data(NULL), type_(type) { create(r, c, type); }
  int type() const { return type_; }
  int channels() const { return (type_ >> 3) + 1; }
  bool empty() const { return data == NULL; }
  void create(int r, int c, int type) {
    if (data && r == rows && c == cols && type == type_) return;
    rows = r; cols = c; type_ = type;
    buf_.reset(new std::vector<unsigned char>(static_cast<size_t>(r) * c * stub_elem_size(type), 0));
    data = buf_->data();
  }
  Mat clone() const {
    Mat m;
    copyTo(m);
    return m;
  }
  void copyTo(Mat& dst) const {
    dst.create(rows, cols, type_);
    if (data) memcpy(dst.data, data, static_cast<size_t>(rows) * cols * stub_elem_size(type_));
  }
  template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[static_cast<size_t>(r) * cols + c]; }
  template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[static_cast<size_t>(r) * cols + c]; }
  // convertTo(dst, rtype, alpha): dst = saturate_cast<rtype>(src * alpha), computed per element in float for a float
  // destination (8U -> 32F with alpha = 1.0f/255.0f is the one the depth filter depends on, depthmap.cpp:105) and rounded
  // half-to-even with saturation for an 8-bit destination (cvRound, scaleMat :166)
  void convertTo(Mat& dst, int rtype, double alpha = 1.0) const {
    const int ddepth = rtype & 7;
    Mat out(rows, cols, CV_MAKETYPE(ddepth, channels()));
    const size_t n = static_cast<size_t>(rows) * cols * channels();
    for (size_t i = 0; i < n; ++i) {
      double v;
      switch (type_ & 7) {
        case CV_8U: v = data[i]; break;
        case CV_32S: v = reinterpret_cast<const int*>(data)[i]; break;
        case CV_32F: v = reinterpret_cast<const float*>(data)[i]; break;
        default: v = reinterpret_cast<const double*>(data)[i]; break;
      }
      if (ddepth == CV_32F) {
        float* o = reinterpret_cast<float*>(out.data);
        o[i] = (type_ & 7) == CV_8U ? static_cast<float>(data[i]) * static_cast<float>(alpha) : static_cast<float>(v * alpha);
      } else if (ddepth == CV_8U) {
        const double s = nearbyint(v * alpha);
        out.data[i] = static_cast<unsigned char>(s < 0 ? 0 : (s > 255 ? 255 : s));
      } else {
        reinterpret_cast<double*>(out.data)[i] = v * alpha;
      }
    }
    dst = out;
  }
 protected:
  int type_;
  std::shared_ptr<std::vector<unsigned char> > buf_;
};

// element-wise arithmetic of scaleMat (depthmap.cpp:158-170): (m - a) * b / c on a float image, in float
inline Mat stub_map(const Mat& m, float (*f)(float, float), double s) {
  Mat out(m.rows, m.cols, m.type());
  const size_t n = static_cast<size_t>(m.rows) * m.cols;
  const float sf = static_cast<float>(s);
  for (size_t i = 0; i < n; ++i) reinterpret_cast<float*>(out.data)[i] = f(reinterpret_cast<const float*>(m.data)[i], sf);
  return out;
}
inline float stub_sub(float a, float b) { return a - b; }
inline float stub_mul(float a, float b) { return a * b; }
inline float stub_div(float a, float b) { return a / b; }
inline Mat operator-(const Mat& m, double s) { return stub_map(m, stub_sub, s); }
inline Mat operator*(const Mat& m, double s) { return stub_map(m, stub_mul, s); }
inline Mat operator/(const Mat& m, double s) { return stub_map(m, stub_div, s); }

template <typename T>
class Mat_ : public Mat {
 public:
  Mat_() : Mat() { type_ = StubType<T>::value; }
  Mat_(int r, int c) : Mat(r, c, StubType<T>::value) {}
  Mat_(int r, int c, T value) : Mat(r, c, StubType<T>::value) {
    for (size_t i = 0; i < static_cast<size_t>(r) * c; ++i) reinterpret_cast<T*>(data)[i] = value;
  }
  Mat_(const Mat& m) : Mat(m) {}
  static Mat_ eye(int r, int c) {
    Mat_ m(r, c);
    for (int i = 0; i < (r < c ? r : c); ++i) m.template at<T>(i, i) = T(1);
    return m;
  }
  // cv::Mat_<float>(3, 3) << a, b, c, ...   (depthmap.cpp:37, :52)
  class Comma {
   public:
    Comma(Mat_ m, T first) : m_(m), i_(0) { put(first); }
    Comma& operator,(T v) { put(v); return *this; }
    operator Mat() const { return m_; }
   private:
    void put(T v) { if (i_ < static_cast<size_t>(m_.rows) * m_.cols) reinterpret_cast<T*>(m_.data)[i_++] = v; }
    Mat_ m_;
    size_t i_;
  };
  Comma operator<<(T first) { return Comma(*this, first); }
};

// m == value on a single-channel int / float image: 8-bit mask, 255 where equal
inline Mat operator==(const Mat& m, double value) {
  Mat mask(m.rows, m.cols, CV_8UC1);
  const size_t n = static_cast<size_t>(m.rows) * m.cols;
  for (size_t i = 0; i < n; ++i) {
    const double v = (m.type() & 7) == CV_32S ? reinterpret_cast<const int*>(m.data)[i] : (m.type() & 7) == CV_32F ? reinterpret_cast<const float*>(m.data)[i] : m.data[i];
    mask.data[i] = v == value ? 255 : 0;
  }
  return mask;
}
inline int countNonZero(const Mat& m) {
  int n = 0;
  for (size_t i = 0; i < static_cast<size_t>(m.rows) * m.cols; ++i) n += m.data[i] != 0;
  return n;
}
// cv::sum of a single-channel float image: accumulated in double, like OpenCV
inline Scalar sum(const Mat& m) {
  double s = 0.0;
  for (size_t i = 0; i < static_cast<size_t>(m.rows) * m.cols; ++i) s += reinterpret_cast<const float*>(m.data)[i];
  return Scalar(s);
}
inline void circle(Mat&, Point, int, const Scalar&, int = 1) {}
inline void line(Mat&, Point, Point, const Scalar&, int = 1) {}

inline void minMaxLoc(const Mat& m, double* min_val, double* max_val) {
  const size_t n = static_cast<size_t>(m.rows) * m.cols;
  double lo = 0, hi = 0;
  for (size_t i = 0; i < n; ++i) {
    const double v = (m.type() & 7) == CV_32F ? reinterpret_cast<const float*>(m.data)[i] : m.data[i];
    if (i == 0 || v < lo) lo = v;
    if (i == 0 || v > hi) hi = v;
  }
  if (min_val) *min_val = lo;
  if (max_val) *max_val = hi;
}

inline void cvtColor(const Mat& src, Mat& dst, int /*CV_GRAY2BGR*/) {
  Mat out(src.rows, src.cols, CV_8UC3);
  const size_t n = static_cast<size_t>(src.rows) * src.cols;
  for (size_t i = 0; i < n; ++i) out.data[3 * i] = out.data[3 * i + 1] = out.data[3 * i + 2] = src.data[i];
  dst = out;
}

// binary PGM (P5, maxval 255) whatever the file is called: what the synthetic dataset exporter writes
inline Mat imread(const std::string& path, int /*flags*/) {
  Mat img;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return img;
  int w = 0, h = 0, maxv = 0;
  char magic[3] = {0, 0, 0};
  if (fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && strcmp(magic, "P5") == 0 && maxv == 255 && w > 0 && h > 0) {
    fgetc(f);  // the single whitespace after maxval
    img.create(h, w, CV_8UC1);
    if (fread(img.data, 1, static_cast<size_t>(w) * h, f) != static_cast<size_t>(w) * h) img = Mat();
  }
  fclose(f);
  return img;
}

inline void imshow(const std::string& window, const Mat& m) {
  const char* dir = getenv("RMD_STUB_IMSHOW_DIR");
  if (!dir) return;
  const std::string path = std::string(dir) + "/" + window + ".bin";
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return;
  const int hdr[3] = {m.rows, m.cols, m.type()};
  fwrite(hdr, sizeof(int), 3, f);
  fwrite(m.data, 1, static_cast<size_t>(m.rows) * m.cols * stub_elem_size(m.type()), f);
  fclose(f);
}
inline int waitKey(int = 0) { return -1; }
inline long long getTickCount() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double getTickFrequency() { return 1e9; }

// cv::initUndistortRectifyMap(K, D, R = I, newK = K, size, CV_16SC2, map1, map2): the library's host restatement
// (rmd_hip_compute_undistortion_map, csrc/rmd_capi.hip)
inline void initUndistortRectifyMap(const Mat& K, const Mat& D, const Mat& /*R*/, const Mat& /*newK*/, Size size, int /*m1type*/, Mat& map1, Mat& map2) {
  map1.create(size.height, size.width, CV_16SC2);
  map2.create(size.height, size.width, CV_MAKETYPE(2 /*CV_16U*/, 1));
  const float* k = reinterpret_cast<const float*>(K.data);
  const float* d = reinterpret_cast<const float*>(D.data);
  rmd_hip_compute_undistortion_map(size.width, size.height, k[0], k[4], k[2], k[5], d[0], d[1], d[2], d[3], reinterpret_cast<short*>(map1.data),
                                   reinterpret_cast<unsigned short*>(map2.data));
}

// cv::remap(src 8UC1, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR), BORDER_CONSTANT 0: OpenCV's fixed-point bilinear
inline void remap(const Mat& src, Mat& dst, const Mat& map1, const Mat& map2, int /*interpolation*/) {
  Mat out(src.rows, src.cols, CV_8UC1);
  const short* m1 = reinterpret_cast<const short*>(map1.data);
  const unsigned short* m2 = reinterpret_cast<const unsigned short*>(map2.data);
  const int w = src.cols, h = src.rows;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const size_t i = static_cast<size_t>(y) * w + x;
      const int sx = m1[2 * i], sy = m1[2 * i + 1], f = m2[i] & 1023, fx = f & 31, fy = f >> 5;
      int v = 0;
      if (!(sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0)) {
        const bool x0 = sx >= 0, x1 = sx + 1 < w, y0 = sy >= 0, y1 = sy + 1 < h;
        const unsigned char* r0 = src.data + static_cast<ptrdiff_t>(sy) * w;
        const unsigned char* r1 = r0 + w;
        const int v00 = (x0 && y0) ? r0[sx] : 0, v01 = (x1 && y0) ? r0[sx + 1] : 0, v10 = (x0 && y1) ? r1[sx] : 0, v11 = (x1 && y1) ? r1[sx + 1] : 0;
        const int sum = v00 * ((32 - fy) * (32 - fx) * 32) + v01 * ((32 - fy) * fx * 32) + v10 * (fy * (32 - fx) * 32) + v11 * (fy * fx * 32);
        v = (sum + (1 << 14)) >> 15;
      }
      out.data[i] = static_cast<unsigned char>(v);
    }
  dst = out;
}

}  // namespace cv
#endif
