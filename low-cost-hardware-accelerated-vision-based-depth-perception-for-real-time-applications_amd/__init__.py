"""MI355X-native ELAS stereo disparity engine (drop-in for the reference's Elas::process hot path).

Sub-modules: engine (ctypes front-end of libstereo_vision_hip.so), build (hipcc driver), synth (seeded test pairs),
stereo_vision (counterpart of the reference's Python entry point).
"""
