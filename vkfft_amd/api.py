"""ctypes binding of the C-ABI in include/vkFFT.h (libvkfft_mi355x.so).

This is the stub a Python caller of the reference (e.g. a pyvkfft-style binding) would use; tests and
bench.py go through it so that everything measured or checked passes the drop-in boundary.  Device
memory comes from torch tensors (ROCm build): plumbing only.  There is no CPU fallback: `load()` raises
if the HIP library is missing.
"""
import ctypes as C
import os

MAXD = 4
u64, i64 = C.c_uint64, C.c_int64
vpp = C.POINTER(C.c_void_p)


class VkFFTConfiguration(C.Structure):
    _fields_ = [
        ("FFTdim", u64), ("size", u64 * MAXD),
        ("device", C.POINTER(C.c_int)), ("stream", vpp), ("num_streams", u64),
        ("userTempBuffer", u64),
        ("bufferNum", u64), ("tempBufferNum", u64), ("inputBufferNum", u64), ("outputBufferNum", u64), ("kernelNum", u64),
        ("bufferSize", C.POINTER(u64)), ("tempBufferSize", C.POINTER(u64)), ("inputBufferSize", C.POINTER(u64)),
        ("outputBufferSize", C.POINTER(u64)), ("kernelSize", C.POINTER(u64)),
        ("buffer", vpp), ("tempBuffer", vpp), ("inputBuffer", vpp), ("outputBuffer", vpp), ("kernel", vpp),
        ("bufferOffset", u64), ("tempBufferOffset", u64), ("inputBufferOffset", u64), ("outputBufferOffset", u64),
        ("kernelOffset", u64), ("specifyOffsetsAtLaunch", u64),
        ("coalescedMemory", u64), ("aimThreads", u64), ("numSharedBanks", u64), ("inverseReturnToInputBuffer", u64),
        ("numberBatches", u64), ("useUint64", u64), ("omitDimension", u64 * MAXD), ("performBandwidthBoost", C.c_int),
        ("groupedBatch", u64 * MAXD),
        ("doublePrecision", u64), ("quadDoubleDoublePrecision", u64), ("quadDoubleDoublePrecisionDoubleMemory", u64),
        ("halfPrecision", u64), ("halfPrecisionMemoryOnly", u64), ("doublePrecisionFloatMemory", u64),
        ("performR2C", u64), ("performDCT", u64), ("performDST", u64), ("disableMergeSequencesR2C", u64),
        ("forceCallbackVersionRealTransforms", u64),
        ("normalize", u64), ("disableReorderFourStep", u64), ("useLUT", i64), ("useLUT_4step", i64),
        ("makeForwardPlanOnly", u64), ("makeInversePlanOnly", u64),
        ("bufferStride", u64 * MAXD), ("isInputFormatted", u64), ("isOutputFormatted", u64),
        ("inputBufferStride", u64 * MAXD), ("outputBufferStride", u64 * MAXD),
        ("swapTo2Stage4Step", u64), ("swapTo3Stage4Step", u64),
        ("considerAllAxesStrided", u64), ("keepShaderCode", u64), ("printMemoryLayout", u64),
        ("saveApplicationToString", u64), ("loadApplicationFromString", u64), ("loadApplicationString", C.c_void_p),
        ("disableSetLocale", u64),
        ("fixMaxRadixBluestein", u64), ("forceBluesteinSequenceSize", u64), ("useCustomBluesteinPaddingPattern", u64),
        ("primeSizes", C.POINTER(u64)), ("paddedSizes", C.POINTER(u64)),
        ("fixMinRaderPrimeMult", u64), ("fixMaxRaderPrimeMult", u64), ("fixMinRaderPrimeFFT", u64), ("fixMaxRaderPrimeFFT", u64),
        ("performZeropadding", u64 * MAXD), ("fft_zeropad_left", u64 * MAXD), ("fft_zeropad_right", u64 * MAXD),
        ("frequencyZeroPadding", u64),
        ("performConvolution", u64), ("conjugateConvolution", u64), ("crossPowerSpectrumNormalization", u64),
        ("coordinateFeatures", u64), ("matrixConvolution", u64), ("symmetricKernel", u64), ("numberKernels", u64),
        ("kernelConvolution", u64),
        ("registerBoost", u64), ("registerBoostNonPow2", u64), ("registerBoost4Step", u64),
        ("devicePageSize", u64), ("localPageSize", u64),
        ("computeCapabilityMajor", u64), ("computeCapabilityMinor", u64),
        ("maxComputeWorkGroupCount", u64 * MAXD), ("maxComputeWorkGroupSize", u64 * MAXD), ("maxThreadsNum", u64),
        ("sharedMemorySizeStatic", u64), ("sharedMemorySize", u64), ("sharedMemorySizePow2", u64), ("warpSize", u64),
        ("halfThreads", u64), ("allocateTempBuffer", u64), ("reorderFourStep", u64), ("maxCodeLength", i64),
        ("maxTempLength", i64), ("autoCustomBluesteinPaddingPattern", u64), ("useRaderUintLUT", u64), ("vendorID", u64),
        ("stream_event", vpp), ("streamCounter", u64), ("streamID", u64), ("useStrict32BitAddress", i64),
    ]


class VkFFTLaunchParams(C.Structure):
    _fields_ = [("buffer", vpp), ("tempBuffer", vpp), ("inputBuffer", vpp), ("outputBuffer", vpp), ("kernel", vpp),
                ("bufferOffset", u64), ("tempBufferOffset", u64), ("inputBufferOffset", u64),
                ("outputBufferOffset", u64), ("kernelOffset", u64)]


class VkFFTPlan(C.Structure):
    _fields_ = [("actualFFTSizePerAxis", (u64 * MAXD) * MAXD), ("numAxisUploads", u64 * MAXD),
                ("axisSplit", (u64 * 4) * MAXD), ("bigSequenceEvenR2C", u64),
                ("actualPerformR2CPerAxis", u64 * MAXD), ("impl", C.c_void_p)]


class VkFFTApplication(C.Structure):
    _fields_ = [
        ("configuration", VkFFTConfiguration),
        ("localFFTPlan", C.POINTER(VkFFTPlan)), ("localFFTPlan_inverse", C.POINTER(VkFFTPlan)),
        ("actualNumBatches", u64), ("firstAxis", u64), ("lastAxis", u64), ("useBluesteinFFT", u64 * MAXD),
        ("bufferRaderUintLUT", (C.c_void_p * 4) * MAXD), ("bufferBluestein", C.c_void_p * MAXD),
        ("bufferBluesteinFFT", C.c_void_p * MAXD), ("bufferBluesteinIFFT", C.c_void_p * MAXD),
        ("bufferRaderUintLUTSize", (u64 * 4) * MAXD), ("bufferBluesteinSize", u64 * MAXD),
        ("applicationBluesteinString", C.c_void_p * MAXD), ("applicationBluesteinStringSize", u64 * MAXD),
        ("numRaderFFTPrimes", u64), ("rader_primes", u64 * 30), ("rader_buffer_size", u64 * 30),
        ("raderFFTkernel", C.c_void_p * 30), ("applicationStringOffsetRader", u64), ("currentApplicationStringPos", u64),
        ("applicationStringSize", u64), ("saveApplicationString", C.c_void_p), ("impl", C.c_void_p),
    ]


EXPORTS = ["initializeVkFFT", "VkFFTAppend", "deleteVkFFT", "VkFFTGetVersion", "getVkFFTErrorString",
           "vkfftMI355XStructSizes", "vkfftMI355XDescribePlan", "vkfftMI355XStreamCopy"]
VKFFT_SUCCESS = 0
_lib = None


def lib_path():
    if os.environ.get("VKFFT_MI355X_LIB"):  # development builds (make dev): tools/ only
        return os.path.abspath(os.environ["VKFFT_MI355X_LIB"])
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvkfft_mi355x.so")


def source_hash():
    """Identifies the build the profiles were taken on: sha1 over the kernel / planner sources (csrc/*, sorted by name).  The .so itself is
    not hashed (46 MB, and a rebuild from the same sources must keep the key)."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def library_is_current():
    """True when the built library is at least as new as every kernel / planner source and generated table: `source_hash()` names the SOURCES a profile was taken
    on, this says that the binary that ran was built from them (bench.py records both)."""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    try:
        so = os.path.getmtime(lib_path())
    except OSError:
        return False
    newest = max(os.path.getmtime(os.path.join(d, n)) for n in os.listdir(d) if n.endswith((".h", ".hip", ".cpp", ".inc")))
    return so >= newest


def _bind(p):
    # torch ships its own copy of the HIP runtime: import it first so that this process ends up with ONE
    # libamdhip64 (the one torch's tensors live in) — device pointers are not shared between two runtimes.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    lib.initializeVkFFT.restype = C.c_int
    lib.initializeVkFFT.argtypes = [C.POINTER(VkFFTApplication), VkFFTConfiguration]
    lib.VkFFTAppend.restype = C.c_int
    lib.VkFFTAppend.argtypes = [C.POINTER(VkFFTApplication), C.c_int, C.POINTER(VkFFTLaunchParams)]
    lib.deleteVkFFT.restype = None
    lib.deleteVkFFT.argtypes = [C.POINTER(VkFFTApplication)]
    lib.VkFFTGetVersion.restype = C.c_int
    lib.getVkFFTErrorString.restype = C.c_char_p
    lib.getVkFFTErrorString.argtypes = [C.c_int]
    lib.vkfftMI355XStructSizes.restype = None
    lib.vkfftMI355XStructSizes.argtypes = [C.POINTER(u64)]
    lib.vkfftMI355XDescribePlan.restype = C.c_int
    lib.vkfftMI355XDescribePlan.argtypes = [C.POINTER(VkFFTApplication), C.c_int, C.c_char_p, u64]
    lib.vkfftMI355XStreamCopy.restype = C.c_int
    lib.vkfftMI355XStreamCopy.argtypes = [C.c_void_p, C.c_void_p, u64, C.c_void_p]
    sizes = (u64 * 4)()
    lib.vkfftMI355XStructSizes(sizes)
    mine = [C.sizeof(VkFFTConfiguration), C.sizeof(VkFFTLaunchParams), C.sizeof(VkFFTPlan), C.sizeof(VkFFTApplication)]
    if list(sizes) != mine:
        raise RuntimeError(f"struct layout mismatch: library {list(sizes)} vs binding {mine}")
    return lib


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` or `make`")
        _lib = _bind(p)
    return _lib


def load_test_double(path):
    """tests only: bind the CPU-emulated build of the same sources (tests/hostemu).  Never used by the product."""
    lib = _bind(path)
    lib._vkfft_test_double = True
    return lib


class VkFFTError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__(f"VkFFT error {code}")


class App:
    """Plan object mirroring the reference call sequence: configure -> initializeVkFFT -> VkFFTAppend -> deleteVkFFT."""

    def __init__(self, size, batch=1, *, dp=False, r2c=False, dct=0, dst=0, normalize=False, device_index=0,
                 buffer_ptr=0, stream=None, streams=None, lib=None, **extra):
        self.lib = lib if lib is not None else load()
        self.cfg = VkFFTConfiguration()
        self.app = VkFFTApplication()
        size = list(size) if hasattr(size, "__len__") else [size]
        self.cfg.FFTdim = len(size)
        for i, s in enumerate(size):
            self.cfg.size[i] = s
        self.cfg.numberBatches = batch
        self.cfg.doublePrecision = int(dp)
        self.cfg.performR2C = int(r2c)
        self.cfg.performDCT = dct
        self.cfg.performDST = dst
        self.cfg.normalize = int(normalize)
        self._dev = C.c_int(device_index)
        self.cfg.device = C.pointer(self._dev)
        self._buf = C.c_void_p(buffer_ptr)
        self.cfg.buffer = C.pointer(self._buf)
        self._keep = []
        if streams is not None:  # several caller streams (VkFFTConfiguration::stream / num_streams)
            self._streams = (C.c_void_p * len(streams))(*streams)
            self.cfg.stream = C.cast(self._streams, type(self.cfg.stream))
            self.cfg.num_streams = len(streams)
        elif stream is not None:
            self._stream = C.c_void_p(stream)
            self.cfg.stream = C.pointer(self._stream)
            self.cfg.num_streams = 1
        for k, v in extra.items():
            cur = getattr(self.cfg, k)
            if hasattr(cur, "__len__"):
                for i, x in enumerate(v):
                    cur[i] = x
            elif isinstance(v, int) and k in ("inputBuffer", "outputBuffer", "tempBuffer", "kernel"):
                slot = C.c_void_p(v)
                self._keep.append(slot)
                setattr(self.cfg, k, C.pointer(slot))
            elif k in ("tempBufferSize", "bufferSize"):
                slot = u64(v)
                self._keep.append(slot)
                setattr(self.cfg, k, C.pointer(slot))
            else:
                setattr(self.cfg, k, v)
        r = self.lib.initializeVkFFT(C.byref(self.app), self.cfg)
        if r != VKFFT_SUCCESS:
            raise VkFFTError(r)
        self.alive = True

    def uploads(self, inverse=False):
        pl = self.app.localFFTPlan_inverse if inverse else self.app.localFFTPlan
        if not pl:  # convolution applications keep their plans in two inner applications; use launch_info()
            return []
        return [int(pl.contents.numAxisUploads[i]) for i in range(int(self.app.configuration.FFTdim))]

    def launch_info(self, inverse=False):
        """(kernel launches per VkFFTAppend, kernel family that accounts for most of them) — extension vkfftMI355XDescribePlan"""
        buf = C.create_string_buffer(1024)
        n = self.lib.vkfftMI355XDescribePlan(C.byref(self.app), 1 if inverse else 0, buf, 1024)
        names = [x for x in buf.value.decode().split(",") if x]
        dom = max(set(names), key=names.count) if names else "?"
        return int(n), dom

    def append(self, inverse, buffer_ptr=None, input_ptr=None, output_ptr=None):
        lp = VkFFTLaunchParams()
        slots = []
        for name, ptr in (("buffer", buffer_ptr), ("inputBuffer", input_ptr), ("outputBuffer", output_ptr)):
            if ptr is not None:
                s = C.c_void_p(ptr)
                slots.append(s)
                setattr(lp, name, C.pointer(s))
        self._last_slots = slots  # the library keeps the pointer-to-pointer (as the reference does)
        r = self.lib.VkFFTAppend(C.byref(self.app), 1 if inverse else -1, C.byref(lp))
        if r != VKFFT_SUCCESS:
            raise VkFFTError(r)

    def forward(self, **kw):
        self.append(False, **kw)

    def inverse(self, **kw):
        self.append(True, **kw)

    def delete(self):
        if getattr(self, "alive", False):
            self.lib.deleteVkFFT(C.byref(self.app))
            self.alive = False

    def __del__(self):
        try:
            self.delete()
        except Exception:
            pass
