// STAND-IN, NOT cv_bridge / OpenCV (syntax check of include/cfear_hip.hpp only): the members of cv_bridge::CvImage and cv::Mat
// that the reference's filter signatures touch (radar_filters.h:88, radar_filters.cpp:40).
#pragma once
#include <boost/shared_ptr.hpp>
#include <cstddef>
#include <cstdint>
#include <vector>
namespace cv {
struct Mat {
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char* data = nullptr;
  std::vector<unsigned char> store;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), step((size_t)c), store((size_t)r * c) { data = store.data(); }
};
}  // namespace cv
namespace cv_bridge {
struct Header { struct Stamp { uint64_t nsec = 0; uint64_t toNSec() const { return nsec; } } stamp; };
struct CvImage { Header header; cv::Mat image; };
typedef boost::shared_ptr<CvImage> CvImagePtr;
}  // namespace cv_bridge
