"""cubicsdr_amd -- MI355X-native implementation of CubicSDR's streaming-IQ DSP hot path.

csrc/        hand-written HIP kernels (gfx950) + the C ABI of include/csdr_hip.h
hip.py       ctypes binding of that ABI
engine.py    thin Python host objects used by tests / bench / smoke
host/        C++ host-side mirror of the reference's IOThread / ThreadBlockingQueue / VisualProcessor surface
"""
__all__ = ["hip", "engine", "build"]
