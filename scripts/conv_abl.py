import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icsg3d_amd import _lib
lib = _lib.load()
abl = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ms = C.c_float(0)
_lib.check(lib.ics_op_conv3d_bench(32, 32, 128, 128, 27, mode, abl, 3, C.byref(ms)))
print("abl", abl, "mode", mode, "ms", ms.value)
