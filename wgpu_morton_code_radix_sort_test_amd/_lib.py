"""ctypes binding of libmrs.so (the C ABI declared in include/mrs.h).

There is no fallback: if the library is missing or fails to load, importing a
function from it raises, so a GPU run can never silently use anything else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRS_LIBRARY_PATH points at another build of the same library (diagnostic builds)
_LIB_PATH = os.environ.get("MRS_LIBRARY_PATH") or os.path.join(_HERE, "csrc", "libmrs.so")

MRS_OK = 0
MRS_ERR_INVALID_ARGUMENT = -1
MRS_ERR_HIP = -2
MRS_ERR_UNSUPPORTED = -3
MRS_ERR_OUT_OF_MEMORY = -4
MRS_ERR_DEVICE_TIMEOUT = -5
MRS_ERR_PEER_FAILED = -6


class MrsError(RuntimeError):
    """A libmrs call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"libmrs error {code}: {message}")
        self.code = code


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("key_words", ctypes.c_int32),
        ("payload_bytes", ctypes.c_int32),
        ("digit_bits", ctypes.c_int32),
        ("begin_bit", ctypes.c_int32),
        ("end_bit", ctypes.c_int32),
        ("algo", ctypes.c_int32),
        ("tile_items", ctypes.c_int32),
    ]


class PlanInfoStruct(ctypes.Structure):
    _fields_ = [
        ("num_passes", ctypes.c_uint32),
        ("radix", ctypes.c_uint32),
        ("tile_items", ctypes.c_uint32),
        ("threads", ctypes.c_uint32),
        ("max_tiles", ctypes.c_uint64),
        ("histogram_len", ctypes.c_uint64),
        ("temp_bytes", ctypes.c_uint64),
    ]


class HybridSplit(ctypes.Structure):
    _fields_ = [
        ("offered", ctypes.c_int32),
        ("windowed", ctypes.c_int32),
        ("leading_bits", ctypes.c_uint32),
        ("shift", ctypes.c_uint32),
        ("base", ctypes.c_uint32),
        ("local_bits", ctypes.c_uint32),
        ("local_capacity", ctypes.c_uint32),
    ]


class DistDesc(ctypes.Structure):
    _fields_ = [
        ("n_local_max", ctypes.c_uint64),
        ("key_words", ctypes.c_int32),
        ("payload_bytes", ctypes.c_int32),
        ("end_bit", ctypes.c_int32),
        ("exchange_chunks", ctypes.c_int32),
        ("tolerance", ctypes.c_uint64),
    ]


class DistInfoStruct(ctypes.Structure):
    _fields_ = [
        ("recv_capacity", ctypes.c_uint64),
        ("device_bytes", ctypes.c_uint64),
        ("last_levels", ctypes.c_uint32),
        ("last_exact_cuts", ctypes.c_uint32),
        ("last_n_out", ctypes.c_uint64),
        ("last_ms", ctypes.c_float * 4),
        ("last_sort_pass_ms", ctypes.c_float),
        ("last_sort_passes", ctypes.c_uint32),
        ("last_sort_schedule", ctypes.c_uint32 * 2),
        ("last_sort_piece_pairs", ctypes.c_uint64),
    ]


_vp = ctypes.c_void_p
_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_i32 = ctypes.c_int32
_fp = ctypes.POINTER(ctypes.c_float)

# name -> (restype, argtypes); every symbol include/mrs.h declares
SIGNATURES = {
    "mrs_last_error_string": (ctypes.c_char_p, []),
    "mrs_version": (ctypes.c_int, []),
    "mrs_context_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "mrs_context_destroy": (ctypes.c_int, [_vp]),
    "mrs_context_get_attribute": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "mrs_context_set_attribute": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64]),
    "mrs_plan_create": (ctypes.c_int, [_vp, ctypes.POINTER(PlanDesc), ctypes.POINTER(_vp)]),
    "mrs_plan_destroy": (ctypes.c_int, [_vp]),
    "mrs_plan_get_info": (ctypes.c_int, [_vp, ctypes.POINTER(PlanInfoStruct)]),
    "mrs_plan_check": (ctypes.c_int, [_vp, _vp]),
    "mrs_plan_get_passes_left_out": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mrs_hybrid_split_for": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.POINTER(HybridSplit)]),
    "mrs_plan_get_schedule": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint32)]),
    "mrs_plan_get_oversized": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.POINTER(ctypes.c_uint32)]),
    "mrs_plan_get_adaptive": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "mrs_plan_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mrs_plan_get_timings": (ctypes.c_int, [_vp, ctypes.c_int, _fp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "mrs_morton_encode_lut": (ctypes.c_int, [_vp, _vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "mrs_morton_encode30": (ctypes.c_int, [_vp, _vp, _u64, _fp, _fp, _vp, _vp, _vp]),
    "mrs_sort_pairs": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mrs_morton_encode30_for_sort": (ctypes.c_int, [_vp, _vp, _u64, _fp, _fp, _vp, _vp, _vp]),
    "mrs_sort_pairs_primed": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mrs_sort_pairs_indexed": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mrs_sort_pairs_noalloc": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mrs_sort_gather": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mrs_histogram": (ctypes.c_int, [_vp, _u64, _vp, _u32, _vp, _vp]),
    "mrs_scan": (ctypes.c_int, [_vp, _u64, _vp, _vp]),
    "mrs_scatter": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "mrs_bucket_histogram": (ctypes.c_int, [_vp, _vp, _u64, _i32, _u32, _u32, _vp, _vp]),
    "mrs_partition": (ctypes.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _vp, _vp]),
    "mrs_comm_get_unique_id": (ctypes.c_int, [_vp]),
    "mrs_comm_create_rccl": (ctypes.c_int, [_vp, _i32, _i32, ctypes.POINTER(_vp)]),
    "mrs_comm_wrap_rccl": (ctypes.c_int, [_vp, _i32, _i32, ctypes.POINTER(_vp)]),
    "mrs_comm_create_local": (ctypes.c_int, [_i32, ctypes.POINTER(_vp)]),
    "mrs_comm_destroy": (ctypes.c_int, [_vp]),
    "mrs_dist_create": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(DistDesc), ctypes.POINTER(_vp)]),
    "mrs_dist_destroy": (ctypes.c_int, [_vp]),
    "mrs_dist_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mrs_dist_get_info": (ctypes.c_int, [_vp, ctypes.POINTER(DistInfoStruct)]),
    "mrs_dist_sort_pairs": (ctypes.c_int, [_vp, _u64, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_u64), _vp]),
    "mrs_dist_check": (ctypes.c_int, [_vp, _vp]),
    "mrs_splitter_create": (ctypes.c_int, [_i32, _i32, _u64, ctypes.POINTER(_vp)]),
    "mrs_splitter_destroy": (ctypes.c_int, [_vp]),
    "mrs_splitter_next": (ctypes.c_int, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u64)]),
    "mrs_splitter_feed": (ctypes.c_int, [_vp, _vp]),
    "mrs_splitter_get": (ctypes.c_int, [_vp, _vp]),
    "mrs_generate_points": (ctypes.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "mrs_generate_u32": (ctypes.c_int, [_vp, _u64, _u64, _u64, _vp, _vp]),
}

_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Loads libmrs.so and binds every declared symbol.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C {os.path.dirname(_LIB_PATH)}` "
            "(or __graft_entry__.build()); there is no non-HIP fallback"
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != MRS_OK:
        msg = load_library().mrs_last_error_string()
        raise MrsError(rc, msg.decode() if msg else "")
