// stand-in: runReductions / finalizeReduction are only used under __CUDACC__
#pragma once
#include "../Platform.hpp"
