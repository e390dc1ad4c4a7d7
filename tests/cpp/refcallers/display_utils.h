// shadows sources/common/display_utils.h (OpenCV mosaics; included by photometric_factor.cpp, unused by it)
#pragma once
