"""detectors/detector_factory.py of the reference: task -> detector class ('semiclass' is outside the hot path)."""
from .tomo_det import TomodetDetector

detector_factory = {"tomo": TomodetDetector, "semi": TomodetDetector, "semi3d": TomodetDetector}
