// <VisionCore/Image/BufferOps.hpp>: included by photometric_factor.cpp, whose only use of it (fillBuffer) is commented out.  TEST INFRASTRUCTURE.
#pragma once
#include <VisionCore/Buffers/Image2D.hpp>
