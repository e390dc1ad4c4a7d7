// The header swap of INTEGRATION.md section 2 (what `#ifdef DFX_DROP_IN` selects in the reference's sources/cuda header of this name):
#pragma once
#include <dfx_shim.hpp>
