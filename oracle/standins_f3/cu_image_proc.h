// Stand-in shadowing the reference header of the same name for oracle/_ref: sparse_geometric_factor.cpp includes it without using it.
#pragma once
