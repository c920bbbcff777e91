// STAND-IN, NOT PCL (syntax check of include/cfear_hip.hpp only).
#pragma once
namespace pcl { struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; }; }
