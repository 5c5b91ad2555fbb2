"""Backend policy object: dtype selection, canonicalisation tolerances, device control.

Mirrors the names a Renormalizer user touches in renormalizer/mps/backend.py:97-216
(``backend.real_dtype``, ``canonical_atol``, ``sync()``, ``free_all_blocks()`` ...).  There
is no array-module alias ``xp`` here: tensors are ``DeviceTensor`` handles owned by the HIP
engine and all arithmetic goes through its C ABI; the functional counterparts of the ``xp.*``
calls the sweep code makes (``tensordot``, ``asnumpy`` / ``asxp``, ``multi_tensor_contract``,
``Matrix``) are in ``renormalizer_amd.mps.matrix``."""
import logging
import os

import numpy as np

from ..engine import DeviceMemoryError, DeviceTensor, get_engine

logger = logging.getLogger("renormalizer_amd")

USE_GPU = True
OE_BACKEND = "mpsengine"     # who contracts the einsum-like expressions (reference: "numpy" / "cupy", backend.py:77-84)
MEMORY_ERRORS = (DeviceMemoryError, MemoryError)
ARRAY_TYPES = (DeviceTensor, np.ndarray)


class Backend:
    _instance_created = False

    def __init__(self):
        if Backend._instance_created:
            raise RuntimeError("Backend should only be initialized once")
        Backend._instance_created = True
        self.first_mp = False
        self._real_dtype = np.float64
        self._complex_dtype = np.complex128
        if os.environ.get("RENO_FP32") is not None:
            raise NotImplementedError("RENO_FP32: the MI355X engine computes in float64/complex128 only")
        self._canonical_atol = 1e-9
        self._canonical_rtol = 1e-5

    @property
    def engine(self):
        return get_engine()

    def free_all_blocks(self):
        self.engine.free_all_blocks()

    def log_memory_usage(self, header=""):
        info = self.engine.mem_info()
        logger.info(f"{header} GPU memory used/pooled: {info['in_use'] / 2**20:.1f}/{info['pool'] / 2**20:.1f} MiB")

    def sync(self):
        self.engine.sync()

    @property
    def is_32bits(self) -> bool:
        return False

    def use_32bits(self):
        raise NotImplementedError("the MI355X engine computes in float64/complex128 only")

    def use_64bits(self):
        pass

    @property
    def real_dtype(self):
        return self._real_dtype

    @property
    def complex_dtype(self):
        return self._complex_dtype

    @property
    def dtypes(self):
        return self._real_dtype, self._complex_dtype

    @property
    def canonical_atol(self):
        return self._canonical_atol

    @canonical_atol.setter
    def canonical_atol(self, value):
        if value < 0:
            raise ValueError("Canonical atol must be non-negative")
        self._canonical_atol = value

    @property
    def canonical_rtol(self):
        return self._canonical_rtol

    @canonical_rtol.setter
    def canonical_rtol(self, value):
        if value < 0:
            raise ValueError("Canonical rtol must be non-negative")
        self._canonical_rtol = value


backend = Backend()
