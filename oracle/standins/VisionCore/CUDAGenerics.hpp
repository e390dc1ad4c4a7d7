// stand-in: the shuffle / reduction generics are only used under __CUDACC__
#pragma once
#include "Platform.hpp"
