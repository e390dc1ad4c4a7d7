#pragma once
#include "se3.hpp"
