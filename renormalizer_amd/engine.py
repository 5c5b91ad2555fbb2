"""ctypes binding of libmpsengine.so (include/mpsengine.h) and device-resident tensors.

This is the only place where Python touches the GPU: no PyTorch, no CuPy.  If the
HIP library is missing or no GPU is visible the engine refuses to start - there is
deliberately no CPU fallback (the NumPy restatement lives in ``oracle/`` and is
test infrastructure only).
"""
import ctypes as C
import math
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RENO_MPSENGINE overrides the in-tree library; there is still no CPU fallback
LIB_PATH = os.environ.get("RENO_MPSENGINE") or os.path.join(_HERE, "csrc", "libmpsengine.so")

F64, C128 = 0, 1
# status codes of include/mpsengine.h
MPSE_OK, MPSE_ERR_OOM, MPSE_ERR_SHAPE, MPSE_ERR_NOCONV, MPSE_ERR_HIP, MPSE_ERR_ARG = 0, 1, 2, 3, 4, 5
DOMAIN_L, DOMAIN_R = 0, 1

STATUS = {0: "OK", 1: "OOM", 2: "SHAPE", 3: "NOCONV", 4: "HIP", 5: "ARG"}


class EngineError(RuntimeError):
    pass


class DeviceMemoryError(EngineError, MemoryError):
    """Device allocation failure (the reference's MEMORY_ERRORS, mps/backend.py:89-94)."""


class mpse_index(C.Structure):
    _fields_ = [("ext", C.c_int64), ("lo_ext", C.c_int64), ("s_hi", C.c_int64), ("s_lo", C.c_int64)]


class mpse_gemm_desc(C.Structure):
    _fields_ = [("dtype_a", C.c_int), ("dtype_b", C.c_int), ("conj_a", C.c_int), ("conj_b", C.c_int),
                ("m_a", mpse_index), ("k_a", mpse_index), ("k_b", mpse_index), ("n_b", mpse_index),
                ("m_c", mpse_index), ("n_c", mpse_index),
                ("batch", C.c_int64), ("sb_a", C.c_int64), ("sb_b", C.c_int64), ("sb_c", C.c_int64),
                ("alpha_re", C.c_double), ("alpha_im", C.c_double), ("beta_re", C.c_double), ("beta_im", C.c_double),
                ("skip_zero_tiles", C.c_int)]


class mpse_dims(C.Structure):
    _fields_ = [("Dl_bra", C.c_int64), ("Dl_ket", C.c_int64), ("Dr_bra", C.c_int64), ("Dr_ket", C.c_int64),
                ("d0", C.c_int64), ("d1", C.c_int64), ("danc", C.c_int64),
                ("wl", C.c_int64), ("wm", C.c_int64), ("wr", C.c_int64), ("env_unit", C.c_int64),
                ("danc1", C.c_int64)]


class mpse_heff(C.Structure):
    _fields_ = [("nsite", C.c_int), ("dims", mpse_dims),
                ("L", C.c_void_p), ("l_dtype", C.c_int),
                ("R", C.c_void_p), ("r_dtype", C.c_int),
                ("W0", C.c_void_p), ("W1", C.c_void_p), ("w_dtype", C.c_int),
                ("l_unit", C.c_int64), ("r_unit", C.c_int64)]


def idx1(ext, stride):
    return mpse_index(int(ext), max(int(ext), 1), 0, int(stride))


def idx2(hi, lo, s_hi, s_lo):
    return mpse_index(int(hi) * int(lo), max(int(lo), 1), int(s_hi), int(s_lo))


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float64:
        return F64
    if dt == np.complex128:
        return C128
    raise TypeError(f"unsupported dtype {dt}; the engine computes in float64 / complex128")


_i64p = C.POINTER(C.c_int64)
_dblp = C.POINTER(C.c_double)

_SIGNATURES = {
    "mpse_ctx_create": [C.c_int, C.POINTER(C.c_void_p)],
    "mpse_ctx_destroy": [C.c_void_p],
    "mpse_sync": [C.c_void_p],
    "mpse_device_info": [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_void_p)],
    "mpse_prof_enable": [C.c_void_p, C.c_int],
    "mpse_prof_reset": [C.c_void_p],
    "mpse_prof_get": [C.c_void_p, C.c_int, _dblp, _dblp, _dblp, C.POINTER(C.c_int64)],
    "mpse_prof_get_ktiles": [C.c_void_p, C.c_int, C.POINTER(C.c_int64)],
    "mpse_prof_get_svd_sweeps": [C.c_void_p, C.POINTER(C.c_int64)],
    "mpse_mpo_site_hint": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64],
    "mpse_block_qr_stats": [C.c_void_p] + [C.POINTER(C.c_int64)] * 3,
    "mpse_heff_fused_stats": [C.c_void_p] + [C.POINTER(C.c_int64)] * 2,
    "mpse_block_qr_optimistic": [C.c_void_p, C.c_int],
    "mpse_block_qr_scheme": [C.c_void_p, C.c_int],
    "mpse_block_qr_check": [C.c_void_p, C.POINTER(C.c_int)],
    "mpse_block_qr_pass_stats": [C.c_void_p] + [C.POINTER(C.c_int64)] * 2,
    "mpse_malloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)],
    "mpse_free": [C.c_void_p, C.c_void_p],
    "mpse_pool_trim": [C.c_void_p],
    "mpse_mem_info": [C.c_void_p] + [C.POINTER(C.c_size_t)] * 4,
    "mpse_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpse_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpse_memcpy_d2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "mpse_memset_zero": [C.c_void_p, C.c_void_p, C.c_size_t],
    "mpse_memcpy_2d": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t],
    "mpse_cast_f64_to_c128": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_conj_inplace": [C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_scal": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_double, C.c_double],
    "mpse_axpy": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double],
    "mpse_mul_real": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_davidson_precond": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                              C.c_double, C.c_double],
    "mpse_real_part": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_dotc": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, _dblp],
    "mpse_nrm2": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, _dblp],
    "mpse_scaled_rms": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, _dblp],
    "mpse_expm_centre_mask": [C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_defer_begin": [C.c_void_p, C.c_int],
    "mpse_defer_end": [C.c_void_p],
    "mpse_defer_arm": [C.c_void_p, C.c_int],
    "mpse_defer_run": [C.c_void_p, C.c_int],
    "mpse_defer_discard": [C.c_void_p],
    "mpse_gemm": [C.c_void_p, C.POINTER(mpse_gemm_desc), C.c_void_p, C.c_void_p, C.c_void_p],
    "mpse_transpose_inner": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int],
    "mpse_env_update": [C.c_void_p, C.c_int, C.c_int, C.POINTER(mpse_dims), C.c_void_p, C.c_int, C.c_void_p,
                        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p],
    "mpse_heff_apply": [C.c_void_p, C.c_int, C.POINTER(mpse_heff), C.c_void_p, C.c_void_p],
    "mpse_heff_apply2": [C.c_void_p, C.c_int, C.POINTER(mpse_heff), C.c_void_p, C.c_void_p],
    "mpse_env_update_multi": [C.c_void_p, C.c_int, C.c_int, C.POINTER(mpse_dims), C.c_int, _i64p, _i64p, C.c_void_p,
                              C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p],
    "mpse_env_unit_channel": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_double, _i64p],
    "mpse_expm_lanczos": [C.c_void_p, C.c_int, C.POINTER(mpse_heff), C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                          C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int)],
    "mpse_davidson": [C.c_void_p, C.c_int, C.POINTER(mpse_heff), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                      C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, _dblp, C.c_void_p,
                      C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mpse_truncate_select": [_dblp, _i64p, C.c_int64, C.c_int64, C.c_double, _i64p, _i64p],
    "mpse_block_qr": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _i64p, _i64p, _i64p, _i64p,
                      C.c_int, C.c_void_p, C.c_void_p, C.c_int64],
    "mpse_block_svd": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _i64p, _i64p, _i64p, _i64p,
                       C.c_void_p, C.c_void_p, _dblp, C.c_int64],
    "mpse_block_svd_full": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _i64p, _i64p, _i64p, _i64p,
                            _i64p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _dblp, C.c_int64],
    "mpse_gather_cols": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, _i64p, _dblp, C.c_int64],
    "mpse_gather_rows": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, _i64p, _dblp, C.c_int64],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["mpse_last_error", "mpse_version"])


def load_library(path=LIB_PATH):
    """dlopen the HIP engine and attach prototypes.  Raises EngineError if it is not built."""
    if not os.path.exists(path):
        raise EngineError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  renormalizer_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mpse_last_error.argtypes = [C.c_void_p]
    lib.mpse_last_error.restype = C.c_char_p
    lib.mpse_version.argtypes = []
    lib.mpse_version.restype = C.c_char_p
    return lib


class _Recording:
    def __init__(self, eng, which):
        self.eng, self.which = eng, which

    def __enter__(self):
        self.eng._check(self.eng.lib.mpse_defer_begin(self.eng.ctx, self.which))
        self.eng.recording_list = self.which
        return self

    def __exit__(self, et, ev, tb):
        self.eng.recording_list = -1
        if et is None:
            self.eng._check(self.eng.lib.mpse_defer_end(self.eng.ctx))
        else:
            self.eng.lib.mpse_defer_discard(self.eng.ctx)
        return False


class _Buffer:
    """Owns one pooled device allocation."""
    __slots__ = ("eng", "ptr", "nbytes", "__weakref__")

    def __init__(self, eng, nbytes):
        self.eng = eng
        self.nbytes = int(nbytes)
        self.ptr = None                     # stays None if the allocation below raises (see __del__)
        p = C.c_void_p()
        eng._check(eng.lib.mpse_malloc(eng.ctx, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        eng = self.eng
        if eng is not None and eng.ctx is not None and self.ptr:
            try:
                eng.lib.mpse_free(eng.ctx, self.ptr)
            except Exception:
                pass
            self.ptr = None


class DeviceTensor:
    """Dense C-ordered float64 / complex128 tensor resident in HBM.

    Plays the role of the reference's ``Matrix`` (mps/matrix.py:13-186) except that the
    data never lives on the host: slicing-free, reshape is a free view, ``to_host``
    is the only PCIe crossing."""
    __slots__ = ("eng", "buf", "offset", "shape", "dtype", "sigmaqn", "unit")

    def __init__(self, eng, buf, offset, shape, dtype):
        self.eng = eng
        self.buf = buf
        self.offset = int(offset)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.sigmaqn = None
        self.unit = 0       # environments only: 1-based MPO channel that is the identity matrix (mpse_env_unit_channel)

    # -- basic properties
    @property
    def ptr(self):
        return self.buf.ptr + self.offset

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return math.prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def code(self):
        return dtype_code(self.dtype)

    @property
    def is_complex(self):
        return self.dtype == np.complex128

    # shape helpers of the reference's Matrix (mps/matrix.py:66-91): a site tensor is (left bond, physical.., right bond)
    @property
    def pdim(self):
        return self.shape[1:-1]

    @property
    def pdim_prod(self):
        return int(np.prod(self.shape[1:-1]))

    @property
    def bond_dim(self):
        return self.shape[0], self.shape[-1]

    @property
    def r_combine_shape(self):
        return self.shape[0], int(np.prod(self.shape[1:]))

    @property
    def l_combine_shape(self):
        return int(np.prod(self.shape[:-1])), self.shape[-1]

    def r_combine(self):
        return self.reshape(self.r_combine_shape)

    def l_combine(self):
        return self.reshape(self.l_combine_shape)

    @property
    def array(self):
        """Host copy (the reference's ``Matrix.array``)."""
        return self.to_host()

    def __repr__(self):
        return f"DeviceTensor(shape={self.shape}, dtype={self.dtype})"

    # -- views / copies
    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        size = math.prod(self.shape)
        if -1 in shape:
            i = shape.index(-1)
            rest = math.prod(s for s in shape if s != -1)
            shape[i] = size // rest if rest else 0
        if math.prod(shape) != size:
            raise ValueError(f"cannot reshape {self.shape} into {tuple(shape)}")
        return DeviceTensor(self.eng, self.buf, self.offset, shape, self.dtype)

    def ravel(self):
        return self.reshape(self.size)

    def row_block(self, start, stop):
        """View of rows [start, stop) along the first axis (contiguous)."""
        inner = math.prod(self.shape[1:]) * self.dtype.itemsize
        return DeviceTensor(self.eng, self.buf, self.offset + start * inner, (stop - start,) + self.shape[1:], self.dtype)

    def shifted(self, nelem):
        """Flat view of the same buffer starting ``nelem`` elements further on (operand offsets of strided GEMMs)."""
        return DeviceTensor(self.eng, self.buf, self.offset + int(nelem) * self.dtype.itemsize,
                            (self.size - int(nelem),), self.dtype)

    def copy(self):
        out = self.eng.empty(self.shape, self.dtype)
        self.eng._check(self.eng.lib.mpse_memcpy_d2d(self.eng.ctx, out.ptr, self.ptr, self.nbytes))
        return out

    def to_complex(self):
        if self.is_complex:
            return self
        out = self.eng.empty(self.shape, np.complex128)
        self.eng._check(self.eng.lib.mpse_cast_f64_to_c128(self.eng.ctx, out.ptr, self.ptr, self.size))
        return out

    def conj(self):
        if not self.is_complex:
            return self
        out = self.copy()
        self.eng._check(self.eng.lib.mpse_conj_inplace(self.eng.ctx, out.ptr, out.size))
        return out

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.size:
            self.eng._check(self.eng.lib.mpse_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    # -- in-place scalar algebra used by the sweeps
    def scale_(self, a):
        a = complex(a)
        if a.imag != 0 and not self.is_complex:
            raise TypeError("complex scale of a real tensor")
        self.eng._check(self.eng.lib.mpse_scal(self.eng.ctx, self.code, self.ptr, self.size, a.real, a.imag))
        return self

    def norm(self):
        out = (C.c_double * 2)()
        self.eng._check(self.eng.lib.mpse_nrm2(self.eng.ctx, self.code, self.ptr, self.size, out))
        return float(out[0])

    def vdot(self, other):
        """sum conj(self) * other"""
        assert other.dtype == self.dtype and other.size == self.size
        out = (C.c_double * 2)()
        self.eng._check(self.eng.lib.mpse_dotc(self.eng.ctx, self.code, self.ptr, other.ptr, self.size, out))
        return complex(out[0], out[1]) if self.is_complex else float(out[0])


class Engine:
    """One HIP context (device + stream + memory pool) per process, as in the
    reference's single-GPU backend (mps/backend.py:129-132)."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("RENO_GPU", "0"))
        self.lib = load_library()
        self.ctx = None
        p = C.c_void_p()
        st = self.lib.mpse_ctx_create(int(device), C.byref(p))
        if st != 0:
            raise EngineError(
                f"mpse_ctx_create(device={device}) failed with status {STATUS.get(st, st)}: no usable MI355X/HIP "
                "device.  renormalizer_amd has no CPU fallback.")
        self.ctx = p.value
        self.device = int(device)
        name = C.create_string_buffer(128)
        ncu = C.c_int()
        stream = C.c_void_p()
        self.lib.mpse_device_info(self.ctx, name, 128, C.byref(ncu), C.byref(stream))
        self.device_name = name.value.decode()
        self.n_cu = ncu.value
        self.stream = stream.value
        self._ones = {}
        self.recording_list = -1            # list being recorded (mpse_defer_begin), -1 = none

    def close(self):
        if self.ctx is not None:
            self.lib.mpse_ctx_destroy(self.ctx)
            self.ctx = None

    # -- status handling
    def _check(self, st):
        if st == 0:
            return
        msg = self.lib.mpse_last_error(self.ctx)
        msg = msg.decode() if msg else ""
        if st == 1:
            raise DeviceMemoryError(f"device out of memory: {msg}")
        if st == 2:
            raise ValueError(f"mpsengine: {msg}")
        raise EngineError(f"mpsengine status {STATUS.get(st, st)}: {msg}")

    def sync(self):
        self._check(self.lib.mpse_sync(self.ctx))

    # -- deferred calls (mpse_defer_*): gemm / block QR / environment updates issued inside ``recording(list)`` are
    # stored and run at the end of the Lanczos solve that follows ``arm(list)``
    def recording(self, which):
        return _Recording(self, which)

    def arm(self, which):
        self._check(self.lib.mpse_defer_arm(self.ctx, which))

    def defer_discard(self):
        self.recording_list = -1
        self.lib.mpse_defer_discard(self.ctx)

    def mem_info(self):
        v = [C.c_size_t() for _ in range(4)]
        self._check(self.lib.mpse_mem_info(self.ctx, *[C.byref(x) for x in v]))
        return dict(pool=v[0].value, in_use=v[1].value, device_free=v[2].value, device_total=v[3].value)

    def free_all_blocks(self):
        self._check(self.lib.mpse_pool_trim(self.ctx))

    def mpo_site_hint(self, dev, host):
        """Describe a real MPO site to the engine (``mpse_mpo_site_hint``): ``dev`` is the device copy of ``host``
        (wl, d, d, wr).  Large one-site matvecs on that site then take the folded plan.  The hint goes with the
        buffer; the device copy must not be written to afterwards."""
        host = np.asarray(host)
        if host.ndim != 4 or np.iscomplexobj(host) or dev.is_complex or dev.offset != 0:
            return
        w = np.ascontiguousarray(host, dtype=np.float64)
        self._check(self.lib.mpse_mpo_site_hint(self.ctx, dev.ptr, w.ctypes.data, *[int(x) for x in w.shape[:2]],
                                                int(w.shape[3])))

    # -- kernel profiling (HIP events on the engine stream)
    def prof_enable(self, on=True):
        """on: False/0 off, True/1 every launch, N > 1 every N-th launch."""
        self._check(self.lib.mpse_prof_enable(self.ctx, int(on)))

    def prof_reset(self):
        self._check(self.lib.mpse_prof_reset(self.ctx))

    def block_qr_stats(self):
        """(block QR calls, of which through the Cholesky-QR kernels, of which redone by Householder) of this context."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.mpse_block_qr_stats(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def heff_fused_stats(self):
        """(bond-matrix, two-level-site) effective-Hamiltonian applications that ran as the fused launch."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.mpse_heff_fused_stats(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def block_qr_optimistic(self, on):
        """Optimistic mode of the Cholesky-QR path (``mpse_block_qr_optimistic``): breakdowns are not read back per
        decomposition but raise a sticky flag - ``block_qr_check()`` at the end of a step that can be repeated."""
        self._check(self.lib.mpse_block_qr_optimistic(self.ctx, int(bool(on))))

    def block_qr_scheme(self, scheme):
        """0 Householder only, 1 Cholesky-QR for tall blocks (default), 2 Cholesky-QR wherever it applies, -1 the
        environment's setting (``mpse_block_qr_scheme``)."""
        self._check(self.lib.mpse_block_qr_scheme(self.ctx, int(scheme)))

    def block_qr_check(self):
        v = C.c_int(0)
        self._check(self.lib.mpse_block_qr_check(self.ctx, C.byref(v)))
        return bool(v.value)

    def block_qr_pass_stats(self):
        """(blocks factorised by the Cholesky-QR kernels, of them finished after two passes) - synchronous."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.mpse_block_qr_pass_stats(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def prof_get(self):
        """{variant: dict(ms, flops, bytes, launches)} for the contraction kernel variants."""
        names = {0: "f64xf64", 1: "c128xf64", 2: "f64xc128", 3: "c128xc128", 4: "lanczos_vec", 5: "block_qr", 6: "block_svd",
                 7: "heff_fused"}
        issued_per_mac = {0: 2.0, 1: 4.0, 2: 4.0, 3: 6.0}     # real flops the MFMA units execute per multiply-add
        out = {}
        for v, nm in names.items():
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            self._check(self.lib.mpse_prof_get(self.ctx, v, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
            out[nm] = dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)
            if v < 4:
                kt = C.c_int64()
                self._check(self.lib.mpse_prof_get_ktiles(self.ctx, v, C.byref(kt)))
                out[nm]["ktiles"] = kt.value
                out[nm]["issued_flops"] = kt.value * 65536.0 * issued_per_mac[v]
        sw = C.c_int64()
        self._check(self.lib.mpse_prof_get_svd_sweeps(self.ctx, C.byref(sw)))
        out["block_svd"]["sweeps"] = sw.value
        return out

    # -- tensor factories
    def empty(self, shape, dtype=np.float64):
        if isinstance(shape, (int, np.integer)):
            shape = (shape,)
        dtype = np.dtype(dtype)
        dtype_code(dtype)
        n = math.prod(int(x) for x in shape)
        buf = _Buffer(self, n * dtype.itemsize)
        return DeviceTensor(self, buf, 0, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        t = self.empty(shape, dtype)
        self._check(self.lib.mpse_memset_zero(self.ctx, t.ptr, t.nbytes))
        return t

    def asdevice(self, a, dtype=None):
        if isinstance(a, DeviceTensor):
            if dtype is not None and np.dtype(dtype) != a.dtype:
                if np.dtype(dtype) == np.complex128:
                    return a.to_complex()
                raise TypeError("cannot cast complex device tensor to real")
            return a
        a = np.asarray(a)
        if dtype is None:
            dtype = np.complex128 if np.iscomplexobj(a) else np.float64
        a = np.ascontiguousarray(a, dtype=dtype)
        t = self.empty(a.shape, a.dtype)
        if a.size:
            self._check(self.lib.mpse_memcpy_h2d(self.ctx, t.ptr, a.ctypes.data, a.nbytes))
        return t

    def copy_block(self, dst, dst_row0, dst_col0, src):
        """dst[dst_row0 + r, dst_col0 + c] = src[r, c] for 2-D views of equal dtype (device to device)."""
        assert dst.dtype == src.dtype and dst.ndim == 2 and src.ndim == 2
        es = dst.dtype.itemsize
        rows, cols = src.shape
        assert dst_row0 + rows <= dst.shape[0] and dst_col0 + cols <= dst.shape[1]
        self._check(self.lib.mpse_memcpy_2d(self.ctx, dst.ptr + (dst_row0 * dst.shape[1] + dst_col0) * es,
                                            dst.shape[1] * es, src.ptr, cols * es, cols * es, rows))

    def copy_sub(self, dst, dst_row0, dst_col0, src, src_row0, src_col0, rows, cols):
        """dst[dst_row0 + r, dst_col0 + c] = src[src_row0 + r, src_col0 + c], r < rows, c < cols (2-D, same dtype)."""
        assert dst.dtype == src.dtype and dst.ndim == 2 and src.ndim == 2
        assert dst_row0 + rows <= dst.shape[0] and dst_col0 + cols <= dst.shape[1]
        assert src_row0 + rows <= src.shape[0] and src_col0 + cols <= src.shape[1]
        if rows == 0 or cols == 0:
            return
        es = dst.dtype.itemsize
        self._check(self.lib.mpse_memcpy_2d(self.ctx, dst.ptr + (dst_row0 * dst.shape[1] + dst_col0) * es,
                                            dst.shape[1] * es, src.ptr + (src_row0 * src.shape[1] + src_col0) * es,
                                            src.shape[1] * es, cols * es, rows))

    def ones(self, shape, dtype=np.float64):
        return self.asdevice(np.ones(shape, dtype=dtype))

    # -- general contraction
    def gemm(self, A, B, Cout, m_a, k_a, k_b, n_b, m_c, n_c, conj_a=False, conj_b=False, batch=1,
             sb_a=0, sb_b=0, sb_c=0, alpha=1.0, beta=0.0, skip_zero_tiles=0):
        d = mpse_gemm_desc()
        d.dtype_a, d.dtype_b = A.code, B.code
        d.conj_a, d.conj_b = int(conj_a), int(conj_b)
        d.m_a, d.k_a, d.k_b, d.n_b, d.m_c, d.n_c = m_a, k_a, k_b, n_b, m_c, n_c
        d.batch, d.sb_a, d.sb_b, d.sb_c = int(batch), int(sb_a), int(sb_b), int(sb_c)
        alpha, beta = complex(alpha), complex(beta)
        d.alpha_re, d.alpha_im, d.beta_re, d.beta_im = alpha.real, alpha.imag, beta.real, beta.imag
        d.skip_zero_tiles = int(skip_zero_tiles)
        self._check(self.lib.mpse_gemm(self.ctx, C.byref(d), A.ptr, B.ptr, Cout.ptr))
        return Cout

    def matmul(self, A, B, conj_a=False, conj_b=False, trans_a=False, trans_b=False):
        """(M,K)@(K,N) on 2-D views; trans_* read the operand transposed through strides."""
        a0, a1 = A.shape
        b0, b1 = B.shape
        M, K = (a1, a0) if trans_a else (a0, a1)
        K2, N = (b1, b0) if trans_b else (b0, b1)
        if K != K2:
            raise ValueError(f"matmul shape mismatch {A.shape} {B.shape}")
        dt = np.complex128 if (A.is_complex or B.is_complex) else np.float64
        out = self.empty((M, N), dt)
        m_a, k_a = (idx1(M, 1), idx1(K, a1)) if trans_a else (idx1(M, a1), idx1(K, 1))
        k_b, n_b = (idx1(K, 1), idx1(N, b1)) if trans_b else (idx1(K, b1), idx1(N, 1))
        return self.gemm(A, B, out, m_a, k_a, k_b, n_b, idx1(M, N), idx1(N, 1), conj_a, conj_b)


_ENGINE = None
_LOCK = threading.Lock()
_TLS = threading.local()


def get_engine() -> Engine:
    """Engine of the calling thread (see ``use_engine``), else the process-wide one, created on first use."""
    eng = getattr(_TLS, "engine", None)
    if eng is not None:
        return eng
    global _ENGINE
    with _LOCK:
        if _ENGINE is None:
            _ENGINE = Engine()
        return _ENGINE


def use_engine(eng):
    """Bind ``eng`` (own HIP stream + memory pool) to the calling thread: several independent trajectories can
    then share one GPU from different threads - ctypes releases the GIL inside the engine, and kernels of
    different streams overlap, so one trajectory's latency-bound phases (QR panels, small solves) hide under
    another's contractions.  Pass None to return to the process-wide engine."""
    _TLS.engine = eng
