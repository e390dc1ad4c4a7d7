#pragma once
#include <VisionCore/Buffers/Image2D.hpp>
