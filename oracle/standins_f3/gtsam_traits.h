// Stand-in shadowing the reference header of the same name for oracle/_ref (gtsam::traits for Sophus types: needs the real GTSAM; unused by linearize()).
#pragma once
