// rmd::test::Dataset of the reference (test/dataset.h, test/dataset.cpp) without OpenCV / Eigen / boost: the sequence file
// (<image name> tx ty tz qx qy qz qw per line, dataset.cpp:100-113), images/<name> (8-bit gray PGM or PNG; colour PNGs are
// converted with OpenCV's fixed-point BGR2GRAY so the bytes match cv::imread(..., GRAYSCALE)), depthmaps/<stem>.depth
// (W*H ASCII floats in centimetres, /100 at dataset.cpp:178).  Header-only; link with -lz.
#ifndef RMD_APPS_DATASET_H
#define RMD_APPS_DATASET_H

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <rmd/se3.cuh>

namespace rmd {
namespace test {

struct GrayImage {
  int width, height;
  std::vector<unsigned char> data;
  GrayImage() : width(0), height(0) {}
};

struct DatasetEntry {
  std::string image_file_name, depthmap_file_name;
  float translation[3];
  float quaternion[4];  // x, y, z, w as in the file
  const std::string& getImageFileName() const { return image_file_name; }
  const std::string& getDepthmapFileName() const { return depthmap_file_name; }
};

namespace detail {

inline bool read_file(const std::string& path, std::vector<unsigned char>& buf) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(n > 0 ? static_cast<size_t>(n) : 0);
  const bool ok = n >= 0 && fread(buf.data(), 1, buf.size(), f) == buf.size();
  fclose(f);
  return ok;
}

inline unsigned char gray_of_rgb(unsigned r, unsigned g, unsigned b) {  // OpenCV 8-bit RGB -> gray
  return static_cast<unsigned char>((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
}

inline bool decode_pgm(const std::vector<unsigned char>& buf, GrayImage& img) {
  if (buf.size() < 2 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '2')) return false;
  const bool binary = buf[1] == '5';
  size_t pos = 2;
  long vals[3];
  for (int k = 0; k < 3;) {
    while (pos < buf.size() && isspace(buf[pos])) ++pos;
    if (pos < buf.size() && buf[pos] == '#') {
      while (pos < buf.size() && buf[pos] != '\n') ++pos;
      continue;
    }
    size_t start = pos;
    while (pos < buf.size() && !isspace(buf[pos])) ++pos;
    if (start == pos) return false;
    vals[k++] = strtol(std::string(buf.begin() + start, buf.begin() + pos).c_str(), NULL, 10);
  }
  const long w = vals[0], h = vals[1], maxval = vals[2];
  if (w <= 0 || h <= 0 || maxval <= 0 || maxval > 65535) return false;
  img.width = static_cast<int>(w); img.height = static_cast<int>(h);
  img.data.resize(static_cast<size_t>(w) * h);
  if (binary) {
    ++pos;  // the single whitespace after maxval
    const size_t bytes = img.data.size() * (maxval < 256 ? 1 : 2);
    if (buf.size() < pos + bytes) return false;
    for (size_t i = 0; i < img.data.size(); ++i) img.data[i] = maxval < 256 ? buf[pos + i] : buf[pos + 2 * i];
  } else {
    std::istringstream ss(std::string(buf.begin() + pos, buf.end()));
    for (size_t i = 0; i < img.data.size(); ++i) {
      long v;
      if (!(ss >> v)) return false;
      img.data[i] = static_cast<unsigned char>(maxval < 256 ? v : v >> 8);
    }
  }
  return true;
}

inline unsigned be32(const unsigned char* p) { return (unsigned(p[0]) << 24) | (unsigned(p[1]) << 16) | (unsigned(p[2]) << 8) | p[3]; }

inline bool decode_png(const std::vector<unsigned char>& buf, GrayImage& img) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat, palette;
  while (pos + 12 <= buf.size()) {
    const unsigned len = be32(&buf[pos]);
    const unsigned char* type = &buf[pos + 4];
    const unsigned char* data = &buf[pos + 8];
    if (pos + 12 + len > buf.size()) return false;
    if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
    else if (!memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!w || !h || interlace || (depth != 8 && depth != 16)) return false;
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels) return false;
  const size_t bpp = channels * depth / 8, stride = static_cast<size_t>(w) * bpp;
  std::vector<unsigned char> raw((stride + 1) * h);
  uLongf raw_len = static_cast<uLongf>(raw.size());
  if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size()) return false;
  std::vector<unsigned char> px(stride * h), zero(stride, 0);
  for (unsigned y = 0; y < h; ++y) {
    const unsigned ft = raw[y * (stride + 1)];
    const unsigned char* line = &raw[y * (stride + 1) + 1];
    unsigned char* cur = &px[y * stride];
    const unsigned char* prev = y ? &px[(y - 1) * stride] : zero.data();
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) {
        const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      } else if (ft != 0) return false;
      cur[i] = static_cast<unsigned char>(line[i] + pred);
    }
  }
  img.width = static_cast<int>(w); img.height = static_cast<int>(h);
  img.data.resize(static_cast<size_t>(w) * h);
  const size_t bps = depth / 8;  // bytes per sample; the high byte comes first
  for (size_t i = 0; i < img.data.size(); ++i) {
    const unsigned char* p = &px[i * bpp];
    if (ctype == 0 || ctype == 4) img.data[i] = p[0];
    else if (ctype == 3) {
      if (3u * p[0] + 2 >= palette.size()) return false;
      img.data[i] = gray_of_rgb(palette[3 * p[0]], palette[3 * p[0] + 1], palette[3 * p[0] + 2]);
    } else img.data[i] = gray_of_rgb(p[0], p[bps], p[2 * bps]);
  }
  return true;
}

}  // namespace detail

class Dataset {
 public:
  Dataset(const std::string& dataset_path, const std::string& sequence_file) : dataset_path_(dataset_path), sequence_file_(sequence_file) {}
  explicit Dataset(const std::string& sequence_file) : sequence_file_(sequence_file) {}

  static const char* getDataPathEnvVar() { return "RMD_TEST_DATA_PATH"; }
  bool loadPathFromEnv() {  // dataset.cpp:198-207
    const char* p = std::getenv(getDataPathEnvVar());
    if (!p) return false;
    dataset_path_ = p;
    return true;
  }
  void setPath(const std::string& p) { dataset_path_ = p; }

  bool readDataSequence(size_t start = 0, size_t end = 0) {  // dataset.cpp:81-127: lines [start, end), end == 0: all
    if (dataset_path_.empty() || sequence_file_.empty()) return false;
    dataset_.clear();
    std::ifstream f((dataset_path_ + "/" + sequence_file_).c_str());
    if (!f.is_open()) return false;
    std::string line;
    for (size_t line_cnt = 0; std::getline(f, line); ++line_cnt) {
      if (line_cnt < start || (end != 0 && line_cnt >= end)) continue;
      std::stringstream ls(line);
      DatasetEntry e;
      ls >> e.image_file_name;
      e.depthmap_file_name = e.image_file_name.substr(0, e.image_file_name.find('.') + 1) + "depth";
      ls >> e.translation[0] >> e.translation[1] >> e.translation[2];
      ls >> e.quaternion[0] >> e.quaternion[1] >> e.quaternion[2] >> e.quaternion[3];
      if (ls.fail()) continue;
      dataset_.push_back(e);
    }
    return true;
  }

  bool readImage(GrayImage& img, const DatasetEntry& entry) const {  // dataset.cpp:129-148
    std::vector<unsigned char> buf;
    if (!detail::read_file(dataset_path_ + "/images/" + entry.image_file_name, buf)) return false;
    return detail::decode_pgm(buf, img) || detail::decode_png(buf, img);
  }

  void readCameraPose(rmd::SE3<float>& pose, const DatasetEntry& e) const {  // dataset.cpp:150-161 -> T_world_curr
    pose = rmd::SE3<float>(e.quaternion[3], e.quaternion[0], e.quaternion[1], e.quaternion[2], e.translation[0], e.translation[1],
                           e.translation[2]);
  }

  bool readDepthmap(std::vector<float>& depthmap, const DatasetEntry& entry, size_t width, size_t height) const {  // :163-186
    std::ifstream f((dataset_path_ + "/depthmaps/" + entry.depthmap_file_name).c_str());
    if (!f.is_open()) return false;
    depthmap.resize(width * height);
    for (size_t i = 0; i < depthmap.size(); ++i) {
      float z;
      if (!(f >> z)) return false;
      depthmap[i] = z / 100.0f;
    }
    return true;
  }

  std::vector<DatasetEntry>::const_iterator begin() const { return dataset_.begin(); }
  std::vector<DatasetEntry>::const_iterator end() const { return dataset_.end(); }
  size_t size() const { return dataset_.size(); }

 private:
  std::string dataset_path_, sequence_file_;
  std::vector<DatasetEntry> dataset_;
};

}  // namespace test
}  // namespace rmd

#endif  // RMD_APPS_DATASET_H
