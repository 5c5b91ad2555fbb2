import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
    try:
        lib = ctypes.CDLL(name)
        print(name, "loaded", [hasattr(lib, f) for f in ("roctxProfilerResume", "roctxProfilerPause", "roctxRangePushA")])
    except OSError as e:
        print(name, "fail", e)
lib = ctypes.CDLL(sys.argv[1])
lib.roctxProfilerResume.argtypes = [ctypes.c_uint64]
lib.roctxProfilerPause.argtypes = [ctypes.c_uint64]
from renormalizer_amd.engine import get_engine
eng = get_engine()
a = eng.asdevice(np.ones((64, 64)))
b = eng.matmul(a, a); eng.sync()
print("resume ->", lib.roctxProfilerResume(0))
for _ in range(5):
    b = eng.matmul(a, a)
eng.sync()
print("pause ->", lib.roctxProfilerPause(0))
b = eng.matmul(a, a); eng.sync()
