"""pypolar_amd -- drop-in for the decode path of ``pypolar`` (ant-uni-bremen/polar-codes) on MI355X.

Mirrors ``pypolar.PolarDecoder`` (reference: python/bindings/decoder_python.cc:17-75): same constructor
arguments, method names, overload behaviour of ``decode_vector`` (float32 first, then int8) and the same
exception types/messages, and adds ``decode_batch`` (B codewords per call, one HIP kernel launch).
``pypolar.PolarEncoder`` (python/bindings/encoder_python.cc:21-61) is mirrored the same way, with
``encode_batch`` beside ``encode_vector``.

All decoding happens in hand-written HIP kernels behind the C ABI of ``include/polar_amd.h``
(``polar-codes_amd/lib/libpolar_amd.so``).  There is no CPU fallback: without the library or without
a GPU, decoding raises.
"""
import ctypes as C
import os

import numpy as np

from .construction import frozen_bits  # noqa: E402,F401  (pypolar.frozen_bits)
from .detector import Detector          # noqa: E402,F401  (pypolar.Detector)

__all__ = ["PolarDecoder", "PolarEncoder", "Puncturer", "Detector", "frozen_bits", "PolarAmdError", "library_path", "load_library", "C_SYMBOLS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYPOLAR_AMD_LIB: load another build of the same library (kernel experiments); never a fallback
_LIB_PATH = os.environ.get("PYPOLAR_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libpolar_amd.so")
_lib = None

PCD_MEM_HOST, PCD_MEM_DEVICE = 0, 1
# None: keep the library's small-batch routing (pcd_small_batch_rows); a number: every new PolarDecoder gets this threshold
SMALL_BATCH_ROWS = None

# every symbol include/polar_amd.h declares
C_SYMBOLS = [
    "pcd_create", "pcd_destroy", "pcd_set_error_detection", "pcd_set_systematic", "pcd_is_systematic",
    "pcd_block_length", "pcd_info_length", "pcd_list_size", "pcd_frozen_bits", "pcd_error_detection_mode",
    "pcd_out_bytes", "pcd_decode_f32", "pcd_decode_i8", "pcd_program_stats", "pcd_launch_info", "pcd_kernel_name",
    "pcd_set_small_batch_rows", "pcd_small_batch_rows",
    "pcd_set_timing", "pcd_kernel_times", "pcd_last_error", "pcd_version",
    "pce_create", "pce_destroy", "pce_set_error_detection", "pce_set_systematic", "pce_is_systematic",
    "pce_block_length", "pce_info_length", "pce_info_bytes", "pce_code_bytes", "pce_frozen_bits",
    "pce_error_detection_mode", "pce_encode",
    "ppu_create", "ppu_destroy", "ppu_block_length", "ppu_parent_block_length", "ppu_output_positions",
    "ppu_puncture", "ppu_depuncture", "ppu_puncture_packed",
]


class PolarAmdError(RuntimeError):
    """Raised for failures that have no counterpart in the reference (no GPU, HIP errors, capacity)."""

    def __init__(self, code, msg):
        super().__init__(f"[pcd {code}] {msg}")
        self.code = code


def library_path():
    return _LIB_PATH


def load_library():
    """Load libpolar_amd.so (fails loudly if it has not been built: ``make -C polar-codes_amd``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: build it with `make -C polar-codes_amd` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(_LIB_PATH)
    vp, u32p = C.c_void_p, C.POINTER(C.c_uint32)
    lib.pcd_create.restype = C.c_int
    lib.pcd_create.argtypes = [C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.pcd_destroy.restype = None
    lib.pcd_destroy.argtypes = [vp]
    lib.pcd_set_error_detection.restype = C.c_int
    lib.pcd_set_error_detection.argtypes = [vp, C.c_uint, C.c_char_p]
    lib.pcd_set_systematic.restype = C.c_int
    lib.pcd_set_systematic.argtypes = [vp, C.c_int]
    lib.pcd_is_systematic.restype = C.c_int
    lib.pcd_is_systematic.argtypes = [vp]
    for f in ("pcd_block_length", "pcd_info_length", "pcd_list_size", "pcd_out_bytes"):
        getattr(lib, f).restype = C.c_size_t
        getattr(lib, f).argtypes = [vp]
    lib.pcd_frozen_bits.restype = C.c_size_t
    lib.pcd_frozen_bits.argtypes = [vp, vp, C.c_size_t]
    lib.pcd_error_detection_mode.restype = C.c_int
    lib.pcd_error_detection_mode.argtypes = [vp, C.c_char_p, C.c_size_t]
    for f in ("pcd_decode_f32", "pcd_decode_i8"):
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.c_int, vp]
    lib.pcd_program_stats.restype = C.c_int
    lib.pcd_program_stats.argtypes = [vp, C.c_int, u32p]
    lib.pcd_launch_info.restype = C.c_int
    lib.pcd_launch_info.argtypes = [vp, C.c_int, u32p]
    lib.pcd_kernel_name.restype = C.c_int
    lib.pcd_kernel_name.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    lib.pcd_set_small_batch_rows.restype = C.c_int
    lib.pcd_set_small_batch_rows.argtypes = [vp, C.c_size_t]
    lib.pcd_small_batch_rows.restype = C.c_size_t
    lib.pcd_small_batch_rows.argtypes = [vp]
    lib.pcd_set_timing.restype = C.c_int
    lib.pcd_set_timing.argtypes = [vp, C.c_int]
    lib.pcd_kernel_times.restype = C.c_int
    lib.pcd_kernel_times.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    if hasattr(lib, "pcd_scl_trace"):  # include/polar_amd_diag.h: the diagnostic build only (PYPOLAR_AMD_LIB=.../libpolar_amd_diag.so)
        lib.pcd_scl_trace.restype = C.c_int
        lib.pcd_scl_trace.argtypes = [vp, vp, vp, C.c_uint]
        lib.pcd_scl_profile.restype = C.c_int
        lib.pcd_scl_profile.argtypes = [vp, vp, vp, vp, vp, C.c_uint, C.POINTER(C.c_uint)]
        lib.pcd_sc_profile.restype = C.c_int
        lib.pcd_sc_profile.argtypes = [vp, vp, vp, vp, vp, C.c_uint, C.POINTER(C.c_uint)]
    lib.pcd_last_error.restype = C.c_char_p
    lib.pcd_version.restype = C.c_char_p
    lib.pce_create.restype = C.c_int
    lib.pce_create.argtypes = [C.c_size_t, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.pce_destroy.restype = None
    lib.pce_destroy.argtypes = [vp]
    lib.pce_set_error_detection.restype = C.c_int
    lib.pce_set_error_detection.argtypes = [vp, C.c_uint, C.c_char_p]
    lib.pce_set_systematic.restype = C.c_int
    lib.pce_set_systematic.argtypes = [vp, C.c_int]
    lib.pce_is_systematic.restype = C.c_int
    lib.pce_is_systematic.argtypes = [vp]
    for f in ("pce_block_length", "pce_info_length", "pce_info_bytes", "pce_code_bytes"):
        getattr(lib, f).restype = C.c_size_t
        getattr(lib, f).argtypes = [vp]
    lib.pce_frozen_bits.restype = C.c_size_t
    lib.pce_frozen_bits.argtypes = [vp, vp, C.c_size_t]
    lib.pce_error_detection_mode.restype = C.c_int
    lib.pce_error_detection_mode.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.pce_encode.restype = C.c_int
    lib.pce_encode.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
    lib.ppu_create.restype = C.c_int
    lib.ppu_create.argtypes = [C.c_size_t, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.ppu_destroy.restype = None
    lib.ppu_destroy.argtypes = [vp]
    for f in ("ppu_block_length", "ppu_parent_block_length"):
        getattr(lib, f).restype = C.c_size_t
        getattr(lib, f).argtypes = [vp]
    lib.ppu_output_positions.restype = C.c_size_t
    lib.ppu_output_positions.argtypes = [vp, vp, C.c_size_t]
    for f in ("ppu_puncture", "ppu_depuncture"):
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [vp, vp, C.c_size_t, vp, C.c_uint, C.c_int, vp]
    lib.ppu_puncture_packed.restype = C.c_int
    lib.ppu_puncture_packed.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
    _lib = lib
    return lib


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system one).  If this library pulled in
    the system runtime first and torch its own copy later, the process would hold two HIP runtimes and the second one
    finds no GPU ("No HIP GPUs are available").  So when a torch wheel with a bundled runtime is installed and not yet
    loaded, its copy is loaded first and libpolar_amd.so binds to it.  PYPOLAR_AMD_SYSTEM_HIP=1 skips this."""
    import sys
    if os.environ.get("PYPOLAR_AMD_SYSTEM_HIP") or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: without it the system runtime is used
        pass


def _raise(code):
    """Translate a pcd_status into the exception pybind11 would surface for the reference."""
    msg = load_library().pcd_last_error().decode()
    if code == -2:      # std::logic_error("Unknown PolarDecoder type!")  -> RuntimeError via pybind11
        raise RuntimeError(msg)
    if code == -4:      # std::invalid_argument from createDecoder          -> ValueError
        raise ValueError(msg)
    if code == -5:      # std::logic_error("CRC INVALID SIZE!")
        raise RuntimeError(msg)
    if code == -6:      # std::runtime_error("Unknown Error detector requested!")
        raise RuntimeError(msg)
    if code == -10:     # std::out_of_range from Puncturer::Puncturer             -> IndexError via pybind11
        raise IndexError(msg)
    if code == -1:
        raise ValueError(msg)
    if code == -3:
        raise NotImplementedError(msg)
    raise PolarAmdError(code, msg)


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _check_device(t, device, what):
    """a CUDA tensor handed over as raw pointers must live on the GPU the handle was created for: the library
    launches on that device, and a pointer of another GPU would fault or silently go through peer access"""
    idx = t.device.index
    if idx is None:
        import torch
        idx = torch.cuda.current_device()
    if idx != device:
        raise ValueError(f"{what} is on cuda:{idx} but this object was created for device {device}")


def _check_out(t, shape, dtype, like, what):
    """caller-supplied output tensors are written through raw pointers: exact shape, dtype, contiguity, device"""
    if not _is_torch(t) or not t.is_cuda:
        raise ValueError(f"{what} must be a CUDA tensor")
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != like.device:
        raise ValueError(f"{what} must be a contiguous {dtype} tensor of shape {tuple(shape)} on {like.device}")


def _foreign_stream(t_list, stream):
    """temporaries made on torch's current stream but consumed on a caller-given stream stay alive for that stream"""
    import torch
    for t in t_list:
        if t is not None:
            t.record_stream(torch.cuda.ExternalStream(stream, device=t.device))


class PolarDecoder:
    """``pypolar.PolarDecoder(blockLength, listSize, frozenBitPositions, decoderType)``."""

    def __init__(self, blockLength, listSize, frozenBitPositions, decoderType, device=0):
        lib = load_library()
        frozen = np.ascontiguousarray(np.asarray(frozenBitPositions, dtype=np.int64).astype(np.uint32))
        self._h = C.c_void_p()
        rc = lib.pcd_create(int(blockLength), int(listSize), frozen.ctypes.data_as(C.c_void_p), frozen.size,
                            str(decoderType).encode(), int(device), C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc)
        self._lib = lib
        self._device = int(device)
        if SMALL_BATCH_ROWS is not None:  # (module-level override of the library's threshold: the test suite sets 0 so that small
            lib.pcd_set_small_batch_rows(self._h, C.c_size_t(int(SMALL_BATCH_ROWS)))  # test batches still run the throughput kernels)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.pcd_destroy(h)
            except Exception:  # interpreter shutdown: the library may already be gone
                pass
            self._h = None

    # ---- accessors (decoder_python.cc:27-33) ---------------------------------------------------
    def blockLength(self):
        return int(self._lib.pcd_block_length(self._h))

    def infoLength(self):
        return int(self._lib.pcd_info_length(self._h))

    def listSize(self):
        return int(self._lib.pcd_list_size(self._h))

    def setSystematic(self, sys):
        self._lib.pcd_set_systematic(self._h, int(bool(sys)))

    def isSystematic(self):
        return bool(self._lib.pcd_is_systematic(self._h))

    def frozenBits(self):
        n = self._lib.pcd_frozen_bits(self._h, None, 0)
        out = np.zeros(n, np.uint32)
        self._lib.pcd_frozen_bits(self._h, out.ctypes.data_as(C.c_void_p), n)
        return [int(x) for x in out]

    def getErrorDetectionMode(self):
        buf = C.create_string_buffer(64)
        self._lib.pcd_error_detection_mode(self._h, buf, 64)
        return buf.value.decode()

    def setErrorDetection(self, size=0, type="crc"):
        rc = self._lib.pcd_set_error_detection(self._h, int(size), str(type).encode())
        if rc:
            _raise(rc)

    # ---- decode_vector (decoder_python.cc:41-74) -----------------------------------------------
    def decode_vector(self, array):
        """One codeword.  int8 ndarray -> the int8 overload, anything else is cast to float32
        (pybind11 tries the float overload first and only the exact dtype skips conversion)."""
        a = np.asarray(array)
        i8 = a.dtype == np.int8
        a = np.ascontiguousarray(a, dtype=np.int8 if i8 else np.float32)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        if a.size != self.blockLength():
            raise RuntimeError("Input vector size != blockSize // 8!")
        out, _, _ = self._decode_host(a.reshape(1, -1), i8, False)
        # the reference returns infoLength()//8 bytes (decoder_python.cc:52)
        return out[0, : self.infoLength() // 8].copy()

    # ---- batched entry point (no reference counterpart) ----------------------------------------
    def decode_batch(self, llr, return_metrics=False, out=None, ok=None, metrics=None, stream=None):
        """Decode B codewords.

        ``llr``: ``[B, N]`` float32 / int8, either a numpy array (host: staged + synchronous) or a
        torch tensor on the decoder's GPU (device pointers, asynchronous on ``stream`` / the current
        torch stream).  Returns ``(bytes[B, ceil(K/8)], ok[B])`` (+ ``metrics[B, L]``).
        """
        if _is_torch(llr):
            return self._decode_torch(llr, return_metrics, out, ok, metrics, stream)
        a = np.asarray(llr)
        i8 = a.dtype == np.int8
        a = np.ascontiguousarray(a, dtype=np.int8 if i8 else np.float32)
        if a.ndim != 2 or a.shape[1] != self.blockLength():
            raise ValueError("llr must have shape [B, blockLength]")
        o, k, m = self._decode_host(a, i8, return_metrics)
        return (o, k.astype(bool), m) if return_metrics else (o, k.astype(bool))

    def _decode_host(self, a, i8, want_metrics):
        B = a.shape[0]
        nb = int(self._lib.pcd_out_bytes(self._h))
        L = max(1, self.listSize())
        out = np.zeros((B, nb), np.uint8)
        ok = np.zeros(B, np.uint8)
        met = np.zeros((B, L), np.float32) if want_metrics else None
        fn = self._lib.pcd_decode_i8 if i8 else self._lib.pcd_decode_f32
        rc = fn(self._h, a.ctypes.data_as(C.c_void_p), B, out.ctypes.data_as(C.c_void_p),
                ok.ctypes.data_as(C.c_void_p), met.ctypes.data_as(C.c_void_p) if want_metrics else None,
                PCD_MEM_HOST, None)
        if rc:
            _raise(rc)
        return out, ok, met

    def _decode_torch(self, llr, want_metrics, out, ok, metrics, stream):
        import torch

        if not llr.is_cuda:
            res = self.decode_batch(llr.numpy(), want_metrics)
            return tuple(torch.from_numpy(np.ascontiguousarray(r)) for r in res)
        _check_device(llr, self._device, "llr")
        src = llr
        i8 = llr.dtype == torch.int8
        if not i8 and llr.dtype != torch.float32:
            llr = llr.float()
        llr = llr.contiguous()
        if llr.dim() != 2 or llr.shape[1] != self.blockLength():
            raise ValueError("llr must have shape [B, blockLength]")
        B = llr.shape[0]
        nb = int(self._lib.pcd_out_bytes(self._h))
        L = max(1, self.listSize())
        if out is None:
            out = torch.empty((B, nb), dtype=torch.uint8, device=llr.device)
        else:
            _check_out(out, (B, nb), torch.uint8, llr, "out")
        if ok is None:
            ok = torch.empty((B,), dtype=torch.uint8, device=llr.device)
        else:
            _check_out(ok, (B,), torch.uint8, llr, "ok")
        if want_metrics and metrics is None:
            metrics = torch.empty((B, L), dtype=torch.float32, device=llr.device)
        elif want_metrics:
            _check_out(metrics, (B, L), torch.float32, llr, "metrics")
        if stream is None:
            stream = torch.cuda.current_stream(llr.device).cuda_stream
        elif llr is not src:
            _foreign_stream([llr], stream)
        fn = self._lib.pcd_decode_i8 if i8 else self._lib.pcd_decode_f32
        rc = fn(self._h, C.c_void_p(llr.data_ptr()), B, C.c_void_p(out.data_ptr()), C.c_void_p(ok.data_ptr()),
                C.c_void_p(metrics.data_ptr()) if want_metrics else None, PCD_MEM_DEVICE, C.c_void_p(stream))
        if rc:
            _raise(rc)
        return (out, ok, metrics) if want_metrics else (out, ok)

    # ---- introspection ---------------------------------------------------------------------------
    def program_stats(self, list_program):
        cnt = (C.c_uint32 * 2)()
        rc = self._lib.pcd_program_stats(self._h, int(bool(list_program)), cnt)
        if rc:
            _raise(rc)
        return int(cnt[0]), int(cnt[1])

    def launch_info(self, list_program):
        info = (C.c_uint32 * 4)()
        rc = self._lib.pcd_launch_info(self._h, int(bool(list_program)), info)
        if rc:
            _raise(rc)
        return {"lds_bytes_per_codeword": int(info[0]), "codewords_per_workgroup": int(info[1]),
                "path_groups": int(info[2]), "workgroups_per_cu_by_lds": int(info[3])}

    def set_small_batch_rows(self, rows):
        """calls of up to `rows` rows run on the low-latency sibling geometry (one codeword per wavefront); 0: never"""
        rc = self._lib.pcd_set_small_batch_rows(self._h, C.c_size_t(int(rows)))
        if rc:
            _raise(rc)

    def small_batch_rows(self):
        return int(self._lib.pcd_small_batch_rows(self._h))

    def kernel_name(self, list_program):
        """name of the kernel decode calls launch for the SC (False) or the list (True) program, as rocprofv3 shows it"""
        buf = C.create_string_buffer(96)
        rc = self._lib.pcd_kernel_name(self._h, int(bool(list_program)), buf, C.c_size_t(len(buf)))
        if rc:
            _raise(rc)
        return buf.value.decode()

    def _need_diag(self):
        if not hasattr(self._lib, "pcd_scl_trace"):
            raise PolarAmdError("diagnostics (trace / cycle profile) exist in the diagnostic build only: make -C polar-codes_amd diag, "
                                "then PYPOLAR_AMD_LIB=<repo>/polar-codes_amd/lib/libpolar_amd_diag.so")

    def scl_trace(self, llr_row, cap=4096):
        """diagnostics: per-leaf (path count, metrics) trace of one row -> float32[n, 1 + L]"""
        a = np.ascontiguousarray(llr_row, dtype=np.float32)
        L = self.listSize()
        tr = np.zeros((cap, 1 + L), np.float32)
        self._need_diag()
        rc = self._lib.pcd_scl_trace(self._h, a.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p), cap)
        if rc:
            _raise(rc)
        return tr

    def scl_profile(self, llr_row, cap=8192):
        """diagnostics: (ops, stages, cycles) per SCL program instruction for one row"""
        a = np.ascontiguousarray(llr_row, dtype=np.float32)
        ops = np.zeros(cap, np.uint8); st = np.zeros(cap, np.uint8); cyc = np.zeros(cap, np.uint32)
        n = C.c_uint()
        self._need_diag()
        rc = self._lib.pcd_scl_profile(self._h, a.ctypes.data_as(C.c_void_p), ops.ctypes.data_as(C.c_void_p),
                                       st.ctypes.data_as(C.c_void_p), cyc.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc:
            _raise(rc)
        k = min(cap, n.value)
        return ops[:k], st[:k], cyc[:k]

    def sc_profile(self, llr_row, cap=8192):
        """diagnostics: (ops, stages, cycles) per SC program instruction for one row"""
        a = np.ascontiguousarray(llr_row, dtype=np.float32)
        ops = np.zeros(cap, np.uint8); st = np.zeros(cap, np.uint8); cyc = np.zeros(cap, np.uint32)
        n = C.c_uint()
        self._need_diag()
        rc = self._lib.pcd_sc_profile(self._h, a.ctypes.data_as(C.c_void_p), ops.ctypes.data_as(C.c_void_p),
                                      st.ctypes.data_as(C.c_void_p), cyc.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc:
            _raise(rc)
        k = min(cap, n.value)
        return ops[:k], st[:k], cyc[:k]

    def set_timing(self, enable=True):
        self._lib.pcd_set_timing(self._h, int(bool(enable)))

    def kernel_times_ms(self):
        """per-call kernel durations (HIP events on the launch stream) since set_timing(True)"""
        n = C.c_size_t()
        rc = self._lib.pcd_kernel_times(self._h, None, 0, C.byref(n))
        if rc:
            _raise(rc)
        ms = np.zeros(max(1, n.value), np.float32)
        rc = self._lib.pcd_kernel_times(self._h, ms.ctypes.data_as(C.c_void_p), ms.size, C.byref(n))
        if rc:
            _raise(rc)
        return ms[: n.value].copy()


class PolarEncoder:
    """``pypolar.PolarEncoder(blockLength, frozenBitPositions)`` (encoder_python.cc:21-26).

    A new encoder is systematic and carries the dummy detector (Encoder::Encoder, encoder.cpp:17-26)."""

    def __init__(self, blockLength, frozenBitPositions, device=0):
        lib = load_library()
        frozen = np.ascontiguousarray(np.asarray(frozenBitPositions, dtype=np.int64).astype(np.uint32))
        self._h = C.c_void_p()
        self._device = int(device)
        rc = lib.pce_create(int(blockLength), frozen.ctypes.data_as(C.c_void_p), frozen.size, int(device),
                            C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc)
        self._lib = lib

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.pce_destroy(h)
            except Exception:  # interpreter shutdown: the library may already be gone
                pass
            self._h = None

    # ---- accessors (encoder_python.cc:27-40) -------------------------------------------------------
    def blockLength(self):
        return int(self._lib.pce_block_length(self._h))

    def infoLength(self):
        return int(self._lib.pce_info_length(self._h))

    def setSystematic(self, sys):
        self._lib.pce_set_systematic(self._h, int(bool(sys)))

    def isSystematic(self):
        return bool(self._lib.pce_is_systematic(self._h))

    def frozenBits(self):
        n = self._lib.pce_frozen_bits(self._h, None, 0)
        out = np.zeros(n, np.uint32)
        self._lib.pce_frozen_bits(self._h, out.ctypes.data_as(C.c_void_p), n)
        return [int(x) for x in out]

    def getErrorDetectionMode(self):
        buf = C.create_string_buffer(64)
        self._lib.pce_error_detection_mode(self._h, buf, 64)
        return buf.value.decode()

    def setErrorDetection(self, size=0, type="crc"):
        rc = self._lib.pce_set_error_detection(self._h, int(size), str(type).encode())
        if rc:
            _raise(rc)

    # ---- encode_vector (encoder_python.cc:41-60) ---------------------------------------------------
    def encode_vector(self, array):
        """One information word: uint8[infoLength // 8] -> packed code word uint8[blockLength // 8].

        Like the reference, a C-contiguous uint8 ndarray is used in place, so the detector's checksum is
        written into the caller's array (Detector::generate works on the buffer it is handed); any other input
        is converted first (pybind11 ``forcecast``) and left untouched."""
        a = np.asarray(array)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        if a.size != self.infoLength() // 8:
            raise RuntimeError("Input vector size != infoSize // 8!")
        if not (a.dtype == np.uint8 and a.flags.c_contiguous and a.flags.writeable):
            a = np.ascontiguousarray(a, dtype=np.uint8).copy()
        nb = int(self._lib.pce_info_bytes(self._h))
        if a.size == nb:
            info = a.reshape(1, nb)
        else:  # K % 8 != 0: the library reads (K + 7) // 8 bytes, the binding only promises K // 8
            info = np.zeros((1, nb), np.uint8)
            info[0, : a.size] = a
        return self._encode_host(info)[0]

    # ---- batched entry point (no reference counterpart) --------------------------------------------
    def encode_batch(self, info, out=None, stream=None):
        """Encode B information words: ``info`` uint8 ``[B, ceil(K/8)]`` (numpy: host, synchronous; torch CUDA
        tensor: device pointers, asynchronous).  The checksum fields of ``info`` are filled in place when it
        is a contiguous uint8 array/tensor.  Returns the packed code words ``[B, N/8]``."""
        nb = int(self._lib.pce_info_bytes(self._h))
        if _is_torch(info):
            import torch

            if not info.is_cuda:
                return torch.from_numpy(self.encode_batch(info.numpy()))
            _check_device(info, self._device, "info")
            src = info
            if info.dtype != torch.uint8 or not info.is_contiguous():
                info = info.to(torch.uint8).contiguous()
            if info.dim() != 2 or info.shape[1] != nb:
                raise ValueError("info must have shape [B, ceil(infoLength / 8)]")
            B = info.shape[0]
            if out is None:
                out = torch.empty((B, self.blockLength() // 8), dtype=torch.uint8, device=info.device)
            else:
                _check_out(out, (B, self.blockLength() // 8), torch.uint8, info, "out")
            if stream is None:
                stream = torch.cuda.current_stream(info.device).cuda_stream
            elif info is not src:
                _foreign_stream([info], stream)
            rc = self._lib.pce_encode(self._h, C.c_void_p(info.data_ptr()), B, C.c_void_p(out.data_ptr()),
                                      PCD_MEM_DEVICE, C.c_void_p(stream))
            if rc:
                _raise(rc)
            return out
        a = np.asarray(info)
        if not (a.dtype == np.uint8 and a.flags.c_contiguous and a.flags.writeable):
            a = np.ascontiguousarray(a, dtype=np.uint8).copy()
        if a.ndim != 2 or a.shape[1] != nb:
            raise ValueError("info must have shape [B, ceil(infoLength / 8)]")
        return self._encode_host(a)

    def _encode_host(self, info):
        B = info.shape[0]
        code = np.zeros((B, self.blockLength() // 8), np.uint8)
        rc = self._lib.pce_encode(self._h, info.ctypes.data_as(C.c_void_p), B, code.ctypes.data_as(C.c_void_p),
                                  PCD_MEM_HOST, None)
        if rc:
            _raise(rc)
        return code


class Puncturer:
    """``pypolar.Puncturer(blockLength, frozenBitPositions)`` (puncturer_python.cc:21-150): rate matching by removing
    the first ``parentBlockLength - blockLength`` frozen positions of the parent code."""

    _DTYPES = (np.float64, np.float32, np.uint8)   # overload order of the binding: double, float, uint8

    def __init__(self, blockLength, frozenBitPositions, device=0):
        lib = load_library()
        frozen = np.ascontiguousarray(np.asarray(frozenBitPositions, dtype=np.int64).astype(np.uint32))
        self._h = C.c_void_p()
        self._device = int(device)
        rc = lib.ppu_create(int(blockLength), frozen.ctypes.data_as(C.c_void_p), frozen.size, int(device), C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc)
        self._lib = lib

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.ppu_destroy(h)
            except Exception:  # interpreter shutdown: the library may already be gone
                pass
            self._h = None

    def blockLength(self):
        return int(self._lib.ppu_block_length(self._h))

    def parentBlockLength(self):
        return int(self._lib.ppu_parent_block_length(self._h))

    def blockOutputPositions(self):
        n = self._lib.ppu_output_positions(self._h, None, 0)
        out = np.zeros(n, np.uint32)
        self._lib.ppu_output_positions(self._h, out.ctypes.data_as(C.c_void_p), n)
        return [int(x) for x in out]

    @classmethod
    def _as_supported(cls, array):
        """pybind11 picks the overload whose dtype matches exactly, otherwise converts to the first one (double)"""
        a = np.asarray(array)
        if a.dtype.type not in cls._DTYPES:
            a = a.astype(np.float64)
        return np.ascontiguousarray(a)

    def _run(self, fn, a, out_cols):
        out = np.zeros((a.shape[0], out_cols), a.dtype)
        rc = fn(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.c_void_p), a.dtype.itemsize,
                PCD_MEM_HOST, None)
        if rc:
            _raise(rc)
        return out

    # ---- one vector (the reference's interface) --------------------------------------------------------
    def puncture(self, array):
        a = self._as_supported(array)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        if a.size != self.parentBlockLength():
            raise RuntimeError("Input vector size != parentBlockSize!")
        return self._run(self._lib.ppu_puncture, a.reshape(1, -1), self.blockLength())[0]

    def depuncture(self, array):
        a = self._as_supported(array)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        if a.size != self.blockLength():
            raise RuntimeError("Input vector size != blockSize!")
        return self._run(self._lib.ppu_depuncture, a.reshape(1, -1), self.parentBlockLength())[0]

    def puncturePacked(self, array):
        a = np.ascontiguousarray(np.asarray(array), dtype=np.uint8)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        if a.size != self.parentBlockLength() // 8:
            raise RuntimeError("Input vector size != parentBlockSize!")
        return self.puncture_packed_batch(a.reshape(1, -1))[0]

    # ---- batched entry points (no reference counterpart): numpy (host) or torch CUDA tensors (device) ----
    def _batch(self, fn, x, in_cols, out_cols, out, stream):
        if _is_torch(x):
            import torch

            if not x.is_cuda:
                return torch.from_numpy(self._batch(fn, x.numpy(), in_cols, out_cols, None, None))
            _check_device(x, self._device, "the input")
            src = x
            if x.dtype not in (torch.float64, torch.float32, torch.uint8):
                x = x.double()
            x = x.contiguous()
            if x.dim() != 2 or x.shape[1] != in_cols:
                raise ValueError(f"expected shape [B, {in_cols}]")
            if out is None:
                out = torch.empty((x.shape[0], out_cols), dtype=x.dtype, device=x.device)
            else:
                _check_out(out, (x.shape[0], out_cols), x.dtype, x, "out")
            if stream is None:
                stream = torch.cuda.current_stream(x.device).cuda_stream
            elif x is not src:
                _foreign_stream([x], stream)
            rc = fn(self._h, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr()), x.element_size(),
                    PCD_MEM_DEVICE, C.c_void_p(stream))
            if rc:
                _raise(rc)
            return out
        a = self._as_supported(x)
        if a.ndim != 2 or a.shape[1] != in_cols:
            raise ValueError(f"expected shape [B, {in_cols}]")
        return self._run(fn, a, out_cols)

    def puncture_batch(self, x, out=None, stream=None):
        """[B, parentBlockLength] -> [B, blockLength] (float64 / float32 / uint8 elements)"""
        return self._batch(self._lib.ppu_puncture, x, self.parentBlockLength(), self.blockLength(), out, stream)

    def depuncture_batch(self, x, out=None, stream=None):
        """[B, blockLength] -> [B, parentBlockLength], punctured positions 0 (e.g. LLRs for PolarDecoder.decode_batch)"""
        return self._batch(self._lib.ppu_depuncture, x, self.blockLength(), self.parentBlockLength(), out, stream)

    def puncture_packed_batch(self, x, out=None, stream=None):
        """packed code words uint8 [B, parentBlockLength / 8] -> [B, blockLength / 8]"""
        nin, nout = self.parentBlockLength() // 8, self.blockLength() // 8
        if _is_torch(x) and x.is_cuda:
            import torch

            _check_device(x, self._device, "the input")
            src = x
            x = x.to(torch.uint8).contiguous()
            if x.dim() != 2 or x.shape[1] != nin:
                raise ValueError(f"expected shape [B, {nin}]")
            if out is None:
                out = torch.empty((x.shape[0], nout), dtype=torch.uint8, device=x.device)
            else:
                _check_out(out, (x.shape[0], nout), torch.uint8, x, "out")
            if stream is None:
                stream = torch.cuda.current_stream(x.device).cuda_stream
            elif x is not src:
                _foreign_stream([x], stream)
            rc = self._lib.ppu_puncture_packed(self._h, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr()),
                                               PCD_MEM_DEVICE, C.c_void_p(stream))
            if rc:
                _raise(rc)
            return out
        a = np.ascontiguousarray(np.asarray(x), dtype=np.uint8)
        if a.ndim != 2 or a.shape[1] != nin:
            raise ValueError(f"expected shape [B, {nin}]")
        o = np.zeros((a.shape[0], nout), np.uint8)
        rc = self._lib.ppu_puncture_packed(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], o.ctypes.data_as(C.c_void_p),
                                           PCD_MEM_HOST, None)
        if rc:
            _raise(rc)
        return o
