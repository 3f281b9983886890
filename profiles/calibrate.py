#!/usr/bin/env python3
"""FETCH_SIZE calibration for the sweep kernel's access shape (MI355X_MICROARCH.md, HBM: 'calibrate on a
known byte count in your own access pattern').  Run under `rocprofv3 --pmc FETCH_SIZE`; prints the bytes the
calibration kernel really read so that factor = bytes / (FETCH_SIZE * 1024) can be formed."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightspinner_amd import _capi
lib = _capi.load_hip_library()
lib.dll.lsx_hip_calibrate_read.argtypes = [C.c_int32, C.c_double, C.c_int32]
lib.dll.lsx_hip_calibrate_read.restype = C.c_double
for seg in (12, 64):
    b = lib.dll.lsx_hip_calibrate_read(0, 2.0, seg)
    print('CALIB seg=%d bytes_read=%.0f' % (seg, b))
