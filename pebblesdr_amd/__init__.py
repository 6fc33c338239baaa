"""pebblesdr_amd -- MI355X-native IQ receive chain behind PebbleSDR's plugin surface.

The product is the HIP library `libpebblegpu.so` (C ABI: include/pebblegpu.h).  This package is the
thin Python face used by the tests and the bench: ctypes bindings plus numpy conveniences.  There is
no CPU implementation here: importing works anywhere, but every compute call needs the built
library and a HIP device and fails loudly otherwise.
"""
from .binding import (  # noqa: F401
    PebbleGpuError, load_library, library_path, ReceiverBank, StreamBank, DeviceBuffer,
    DM_AM, DM_SAM, DM_FMN, DM_FMM, DM_FMS, DM_DSB, DM_LSB, DM_USB, DM_CWL, DM_CWU, DM_DIGL, DM_DIGU, DM_NONE,
)
from .steps import Mixer, Decimator, DownConvert, FastFIR, Demod, Spectrum  # noqa: F401

__all__ = [
    "PebbleGpuError", "load_library", "library_path", "ReceiverBank", "StreamBank", "DeviceBuffer",
    "Mixer", "Decimator", "DownConvert", "FastFIR", "Demod", "Spectrum",
]
