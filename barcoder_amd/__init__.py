"""barcoder_amd — MI355X-native barcode / guide matcher.

The hot path (read -> library matching, PAM site scan) runs in hand-written HIP
kernels behind the C ABI of include/barcoder.h; this package is the host-side
mirror of the reference's Python interfaces for that path."""
from ._native import BarcoderNativeError, device_count  # noqa: F401
from .library import BarCodeLibrary, BarCodeLibraryError, BarCodeLibraryReader, DeviceLibrary  # noqa: F401
from .matcher import GpuMatcher, MatchResult, MultiGpuMatcher  # noqa: F401

__version__ = "0.1.0"
