// stand-in: dense_sfm.h includes this header and uses nothing from it
#pragma once
