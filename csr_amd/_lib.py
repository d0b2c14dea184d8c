"""
ctypes binding of libcsrk.so (include/csrk.h).  This is the ONLY way the package computes:
there is no CPU fallback.  If the library is missing the import fails loudly; if there is no
GPU every compute call raises CsrkError carrying the HIP error text.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSRK_LIBRARY: load another build of the same sources (kernel experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get('CSRK_LIBRARY') or os.path.join(_HERE, 'libcsrk.so')

OK = 0
ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_OVERFLOW = -1, -2, -3, -4
VAL_NONE, VAL_F32, VAL_F64 = 0, 1, 2
SPMV_AUTO, SPMV_MERGE, SPMV_VECTOR, SPMV_SCALAR = 0, 1, 2, 3


class CsrkError(RuntimeError):
    "A libcsrk call failed (message from csrk_last_error())."

    def __init__(self, code, msg):
        super().__init__(f'libcsrk error {code}: {msg}')
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is not built: run `python csr_amd/build.py` (needs hipcc). '
        'csr_amd has no CPU fallback.')

# One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as
# /opt/rocm's).  If torch initialises its copy AFTER libcsrk has pulled in the system copy, the
# process ends up with two runtimes and the second one to touch the GPU sees no device.  Loading
# torch's copy first makes the dynamic linker resolve libcsrk's DT_NEEDED to it.  torch is optional:
# without it libcsrk simply uses /opt/rocm's runtime.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

lib = C.CDLL(LIB_PATH)

handle_t = C.c_ssize_t     # intptr_t
_vp = C.c_void_p
_i32, _i64, _int = C.c_int32, C.c_int64, C.c_int

# name -> (restype, argtypes); mirrors include/csrk.h one to one
SIGNATURES = {
    'csrk_version': (_int, []),
    'csrk_last_error': (C.c_char_p, []),
    'csrk_device_count': (_int, [C.POINTER(_int)]),
    'csrk_set_device': (_int, [_int]),
    'csrk_synchronize': (_int, [_vp]),
    'csrk_partition_rows': (_int, [C.c_int32, _vp, _int, C.c_int32, C.POINTER(C.c_int32)]),
    'csrk_trim_cache': (_int, []),
    'csrk_create': (_int, [_i32, _i32, _i64, _vp, _int, _vp, _vp, _int, C.POINTER(handle_t)]),
    'csrk_create_device': (_int, [_i32, _i32, _i64, _vp, _int, _vp, _vp, _int, C.POINTER(handle_t)]),
    'csrk_free': (_int, [handle_t]),
    'csrk_device_bytes': (_int, [handle_t, C.POINTER(_i64)]),
    'csrk_info': (_int, [handle_t, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64),
                         C.POINTER(_int), C.POINTER(_int)]),
    'csrk_export': (_int, [handle_t, _vp, _vp, _vp]),
    'csrk_device_ptrs': (_int, [handle_t, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    'csrk_spmv': (_int, [handle_t, _vp, _vp]),
    'csrk_spmv_f32x': (_int, [handle_t, _vp, _vp]),
    'csrk_spmv_f32x_device': (_int, [handle_t, _vp, _vp, _vp]),
    'csrk_spmv_device_part': (_int, [handle_t, _vp, _vp, _vp, _int]),
    'csrk_spmv_cut_rows': (_int, [handle_t, _vp, _i64, C.POINTER(_i64)]),
    'csrk_spmv_device': (_int, [handle_t, _vp, _vp, _vp]),
    'csrk_set_spmv_algo': (_int, [handle_t, _int]),
    'csrk_spmv_algo_name': (C.c_char_p, [handle_t]),
    'csrk_spmv_plan_info': (_int, [handle_t, C.POINTER(_i64), C.POINTER(_i32)]),
    'csrk_spmv_plan_stats': (_int, [handle_t, C.POINTER(_i64), _int]),
    'csrk_spmv_profile_begin': (_int, [handle_t, _int]),
    'csrk_spmv_profile_every': (_int, [handle_t, _int]),
    'csrk_spmv_profile_channels': (_int, [handle_t, _int]),
    'csrk_spmv_profile_end': (_int, [handle_t, C.POINTER(_int), C.POINTER(C.c_float)]),
    'csrk_spmv_profile_end4': (_int, [handle_t, C.POINTER(_int), C.POINTER(C.c_float)]),
    'csrk_spgemm_ab': (_int, [handle_t, handle_t, C.POINTER(handle_t)]),
    'csrk_spgemm_abt': (_int, [handle_t, handle_t, C.POINTER(handle_t)]),
    'csrk_spgemm_set_order': (_int, [_int]),
    'csrk_spgemm_get_order': (_int, [C.POINTER(_int)]),
    'csrk_spgemm_last_route': (_int, [C.POINTER(_int)]),
    'csrk_spmm_dense': (_int, [handle_t, _vp, _i32, _i64, _vp, _i64]),
    'csrk_spmm_dense_device': (_int, [handle_t, _vp, _i32, _i64, _vp, _i64, _vp]),
    'csrk_spmm_plan_stats': (_int, [handle_t, _vp, _int]),
    'csrk_from_coo': (_int, [_i32, _i32, _i64, _vp, _vp, _vp, _int, C.POINTER(handle_t)]),
    'csrk_transpose': (_int, [handle_t, _int, C.POINTER(handle_t)]),
    'csrk_row_nnzs': (_int, [handle_t, _vp]),
    'csrk_row_extent': (_int, [handle_t, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    'csrk_unit_rows': (_int, [handle_t, _vp]),
    'csrk_unit_rows_device': (_int, [handle_t, _vp]),
    'csrk_center_rows_device': (_int, [handle_t, _vp]),
    'csrk_center_rows': (_int, [handle_t, _vp]),
    'csrk_order_columns': (_int, [handle_t]),
    'csrk_filter_zeros': (_int, [handle_t, C.POINTER(handle_t)]),
    'csrk_pick_rows': (_int, [handle_t, C.c_void_p, C.c_int64, C.c_int, C.POINTER(handle_t)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)       # AttributeError here = header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.csrk_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != OK:
        msg = last_error()
        if rc == ERR_INVALID:
            raise ValueError(f'libcsrk: {msg}')
        raise CsrkError(rc, msg)


def ptr(a):
    "data pointer of a numpy array (or None)"
    return None if a is None else a.ctypes.data_as(_vp)
