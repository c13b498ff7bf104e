"""linemod_pose_estimation_amd -- MI355X-native LINEMOD template matching behind the reference's
rgbdDetector::linemod_detection / cv::linemod::Detector::match call surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI -> liblmx.so), the host-side mirror of the
reference interface (detector.py), the bank container (bank.py), the synthetic bank/scene generator the tests
and bench use (synth.py) and the template-shard helper for multi-GPU runs (dist.py).
"""
from .bank import TemplateBank  # noqa: F401
from .detector import Detector, NativeBank, PinnedArena, linemod_detection, merge_raw, MATCH_DTYPE, RAW_MATCH_DTYPE  # noqa: F401
