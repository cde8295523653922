// test stub: the slice of roscpp that src/main_ros.cpp, src/depthmap_node.cpp and src/publisher.cpp use, without ROS.
// "Topics" are files: every publish() writes <RMD_STUB_TOPIC_DIR>/<topic, '/' -> '_'>.<n>.bin and appends a line
// "<messages delivered so far> <topic> <n>" to <RMD_STUB_TOPIC_DIR>/events.txt.  The "bag" that feeds the subscriber is
// $RMD_STUB_BAG: int32 n, width, height; then n x { float64 position[3], float64 orientation wxyz[4], float32 min_depth,
// max_depth, uint8 image[width * height] }.  ros::ok() is true until every message has been delivered by ros::spinOnce().
// Parameters ($RMD_STUB_PARAMS, lines "name value") are read by ros::init.
#ifndef RMD_TEST_STUB_ROS
#define RMD_TEST_STUB_ROS
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#define ROS_ERROR(...) do { fprintf(stderr, "[ROS_ERROR] "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } while (0)
#define ROS_INFO(...) do { printf(__VA_ARGS__); printf("\n"); } while (0)

namespace ros {

namespace stub {
struct BagMessage {
  double position[3], orientation_wxyz[4];
  float min_depth, max_depth;
  int width, height;
  std::vector<unsigned char> image;
};
struct World {
  std::map<std::string, std::string> params;
  std::vector<BagMessage> bag;
  size_t delivered = 0;
  std::function<void(const BagMessage&)> subscriber;
  std::map<std::string, int> topic_seq;
  std::string topic_dir;
};
inline World& world() {
  static World w;
  return w;
}
// opens the next file of a topic and logs the publication
inline FILE* open_topic(const std::string& topic) {
  World& w = world();
  if (w.topic_dir.empty()) return NULL;
  std::string name = topic;
  for (size_t i = 0; i < name.size(); ++i)
    if (name[i] == '/') name[i] = '_';
  const int n = w.topic_seq[name]++;
  std::ofstream(w.topic_dir + "/events.txt", std::ios::app) << w.delivered << " " << name << " " << n << "\n";
  std::ostringstream path;
  path << w.topic_dir << "/" << name << "." << n << ".bin";
  return fopen(path.str().c_str(), "wb");
}
}  // namespace stub

struct Time {
  double sec;
  Time() : sec(0.0) {}
  static Time now() {
    Time t;
    t.sec = static_cast<double>(stub::world().delivered);
    return t;
  }
};

inline void init(int&, char**, const std::string&) {
  stub::World& w = stub::world();
  if (const char* p = getenv("RMD_STUB_PARAMS")) {
    std::ifstream f(p);
    std::string name, value;
    while (f >> name >> value) w.params[name] = value;
  }
  if (const char* d = getenv("RMD_STUB_TOPIC_DIR")) w.topic_dir = d;
  if (const char* b = getenv("RMD_STUB_BAG")) {
    FILE* f = fopen(b, "rb");
    int hdr[3] = {0, 0, 0};
    if (f && fread(hdr, sizeof(int), 3, f) == 3) {
      for (int k = 0; k < hdr[0]; ++k) {
        stub::BagMessage m;
        m.width = hdr[1]; m.height = hdr[2];
        m.image.resize(static_cast<size_t>(hdr[1]) * hdr[2]);
        if (fread(m.position, sizeof(double), 3, f) != 3 || fread(m.orientation_wxyz, sizeof(double), 4, f) != 4 || fread(&m.min_depth, sizeof(float), 1, f) != 1 ||
            fread(&m.max_depth, sizeof(float), 1, f) != 1 || fread(m.image.data(), 1, m.image.size(), f) != m.image.size())
          break;
        w.bag.push_back(m);
      }
    }
    if (f) fclose(f);
  }
}
inline bool ok() { return stub::world().delivered < stub::world().bag.size(); }
inline void spinOnce() {
  stub::World& w = stub::world();
  if (w.delivered < w.bag.size() && w.subscriber) {
    const stub::BagMessage& m = w.bag[w.delivered++];
    w.subscriber(m);
  }
}
struct Rate {
  explicit Rate(double) {}
  void sleep() {}
};

class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string& topic) : topic_(topic) {}
  template <class M> void publish(const std::shared_ptr<M>& msg) const { publish(*msg); }
  template <class M> void publish(const M& msg) const {
    FILE* f = stub::open_topic(topic_);
    if (!f) return;
    msg.stubWrite(f);
    fclose(f);
  }
 private:
  std::string topic_;
};
struct Subscriber {};

class NodeHandle {
 public:
  bool ok() const { return true; }
  template <class M> Publisher advertise(const std::string& topic, uint32_t) { return Publisher(topic); }
  template <class M, class T>
  Subscriber subscribe(const std::string&, uint32_t, void (T::*fp)(const std::shared_ptr<M const>&), T* obj) {
    stub::world().subscriber = [fp, obj](const stub::BagMessage& b) {
      std::shared_ptr<M> m(new M);
      m->stubFill(b);
      (obj->*fp)(m);
    };
    return Subscriber();
  }
};

}  // namespace ros

namespace std_msgs {
struct Header {
  std::string frame_id;
  ros::Time stamp;
};
}  // namespace std_msgs
#endif
