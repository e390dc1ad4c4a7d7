// shadows sources/common/nearest_psd.h (included by photometric_factor.cpp; its only use is commented out)
#pragma once
