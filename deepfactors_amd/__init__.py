"""deepfactors_amd -- MI355X-native (gfx950, HIP) dense photometric alignment kernels that replace the CUDA hot
path of jczarnowski/DeepFactors (SE3Aligner / SfmAligner / UpdateDepth and their image-proc siblings) behind the
reference's own operator interface.  The product is libdfx.so (C ABI, include/dfx.h); this package is the thin
host-side mirror of the reference interface plus the device-memory/stream plumbing (PyTorch)."""
from ._lib import DfxError, EXPORTED_SYMBOLS, LIB_PATH, item_size  # noqa: F401
from .aligners import (BuildPyramids, CameraTracker, Context, CorrespondenceReductionItem, DenseSfmParams, DepthAligner, DeviceImage, GaussianBlurDown,  # noqa: F401
                       JTJJrReductionItem, SE3Aligner, SfmAligner, SfmAlignerParams, SobelGradients, SparseGeometricFactor, SquaredError,
                       TrackerConfig, UpdateDepth, UpdateDepthBatch, default_context, make_pyramids)
from .keyframe import (Frame, Keyframe, KeyframeMap, LoadJsonNetworkConfig, NetworkConfig, save_keyframes, save_results,  # noqa: F401
                       save_trajectory_tum, write_png)
from .factors import HessianBlocks, PhotometricFactor, linearize_all, pose_equals, pose_local  # noqa: F401

__version__ = "0.1.0"
