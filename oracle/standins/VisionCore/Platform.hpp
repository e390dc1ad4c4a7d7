// Stand-in for VisionCore/Platform.hpp (jczarnowski/vision_core, absent): only the function-attribute macros the reference's
// headers use on the host.  TEST INFRASTRUCTURE (oracle/_ref).
#pragma once
#include <Eigen/Core>
#define EIGEN_PURE_DEVICE_FUNC
