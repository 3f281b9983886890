"""TEST INFRASTRUCTURE -- scalar C restatement of the reference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  lightspinner_amd (the product) never does."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'liblsx_oracle.so')


def build(force=False):
    src = os.path.join(HERE, 'lsx_oracle.c')
    hdr = os.path.join(HERE, '..', 'include', 'lsx.h')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(['make', '-s', '-C', HERE, 'liblsx_oracle.so'])
    return LIB


def load():
    """-> lightspinner_amd._capi.LsxLibrary bound to the oracle (same ABI as the HIP library)."""
    import ctypes as C
    from lightspinner_amd._capi import LsxLibrary, _dp
    lib = LsxLibrary(build())
    assert lib.backend == 'oracle-c'
    d = lib.dll
    d.lsx_oracle_w2.argtypes = [C.c_double, _dp]
    d.lsx_oracle_w2.restype = None
    d.lsx_oracle_piecewise_1d_impl.argtypes = [C.c_double, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp, _dp, _dp]
    d.lsx_oracle_piecewise_1d_impl.restype = None
    d.lsx_oracle_planck.argtypes = [C.c_double, C.c_double]
    d.lsx_oracle_planck.restype = C.c_double
    d.lsx_oracle_set_threads.argtypes = [C.c_void_p, C.c_int32]
    return lib
