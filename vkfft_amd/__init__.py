"""MI355X-native FFT library behind the VkFFT C API.  The product is vkfft_amd/lib/libvkfft_mi355x.so
(C-ABI in include/vkFFT.h); this package only carries its sources (csrc/) and the ctypes binding used by
the tests and the benchmark."""
from . import api  # noqa: F401
