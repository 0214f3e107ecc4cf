"""Counterpart of the reference's ctypes wrapper (reference: stereo_vision/sv.py:156-192) over libstereo_vision_hip.so.
(Completed together with csrc/legacy.cpp.)"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_STEREO_VISION_SO_PATH = os.path.join(os.path.dirname(HERE), "libstereo_vision_hip.so")
