// STAND-IN, NOT PCL (syntax check of include/cfear_hip.hpp only).
#pragma once
#include <boost/shared_ptr.hpp>
#include <vector>
namespace pcl {
template <class T> struct PointCloud {
  typedef boost::shared_ptr<PointCloud<T>> Ptr;
  std::vector<T> points;
  void push_back(const T& p) { points.push_back(p); }
  size_t size() const { return points.size(); }
};
}  // namespace pcl
