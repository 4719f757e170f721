"""stylish_tts_amd: MI355X (gfx950) native acoustic hot path of stylish-tts.

Python host layer over libstylish_hip.so (C ABI in include/stylish_hip.h).  There is no CPU or PyTorch
fallback: importing the library without the built .so, or calling a compute entry point without a HIP
device, raises.
"""
from .lib import LIB, StyError, load  # noqa: F401
from .modules import (DurationPredictor, DurationProcessor, ExportModel, MelStyleEncoder, MultiGenerator,  # noqa: F401
                      PitchEnergyPredictor, PitchStyleEncoder, SpeechPredictor)
