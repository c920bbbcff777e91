// STAND-IN, NOT BOOST: boost::shared_ptr as an alias of std::shared_ptr (syntax check of include/cfear_hip.hpp only).
#pragma once
#include <memory>
namespace boost { template <class T> using shared_ptr = std::shared_ptr<T>; }
