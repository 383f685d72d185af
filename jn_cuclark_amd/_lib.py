"""ctypes binding of libmcclark.so (include/mc_api.h).  Fails loudly when the HIP
library is missing: there is no CPU fallback anywhere in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MC_F_FINAL = 1
MC_F_ROWS = 2
MC_FINAL_ROW = 5


class McError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmcclark error %d: %s" % (code, msg))
        self.code = code


class McDbInfo(C.Structure):
    _fields_ = [("htsize", C.c_uint64), ("shard_begin", C.c_uint64), ("shard_end", C.c_uint64),
                ("n_keys", C.c_uint64), ("n_overflow_buckets", C.c_uint64),
                ("n_overflow_keys", C.c_uint64), ("line_bytes", C.c_uint32),
                ("line_capacity", C.c_uint32), ("device_bytes", C.c_uint64),
                ("index_kind", C.c_uint32), ("index_fallback", C.c_uint32),
                ("part", C.c_uint32), ("n_parts", C.c_uint32), ("n_keys_owned", C.c_uint64),
                ("n_lines", C.c_uint64), ("line_begin", C.c_uint64), ("line_end", C.c_uint64),
                ("n_extra_lines", C.c_uint64), ("n_lines_crowded", C.c_uint64),
                ("n_lines_overflowing", C.c_uint64), ("n_spilled_keys", C.c_uint64),
                ("largest_line", C.c_uint32), ("reserved_", C.c_uint32)]


MC_INDEX_BUCKET_LINES, MC_INDEX_MINIMIZER = 0, 1


class McIndexPlan(C.Structure):
    _fields_ = [("fill", C.c_double), ("lines_per_part", C.c_uint64), ("bytes_per_part", C.c_uint64),
                ("fits", C.c_uint32), ("min_parts", C.c_uint32)]


class McStats(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("reads_over_maxhits", C.c_uint64),
                ("kernel_launches", C.c_uint64)]


def library_path():
    # MC_LIB_PATH: A/B builds of the same ABI (kernel tuning); default = the in-tree build
    return os.environ.get("MC_LIB_PATH") or os.path.join(_HERE, "libmcclark.so")


# every symbol include/mc_api.h declares: (name, restype, argtypes)
_vp, _u64, _u32, _i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
SYMBOLS = [
    ("mc_last_error", C.c_char_p, []),
    ("mc_api_version", _i, []),
    ("mc_device_count", _i, [C.POINTER(_i)]),
    ("mc_open", _i, [C.POINTER(_vp), _i, _u32, _u64, _u32, _u32]),
    ("mc_close", _i, [_vp]),
    ("mc_load_db", _i, [_vp, C.c_char_p, _i, _u32, _u64, _u64]),
    ("mc_load_db_host", _i, [_vp, _vp, _vp, _i, _vp, _u64, _u64, _u64]),
    ("mc_load_db_device", _i, [_vp, _vp, _vp, _i, _vp, _u64, _u64, _u64]),
    ("mc_load_db_part", _i, [_vp, C.c_char_p, _i, _u32, _u32, _u32]),
    ("mc_index_begin", _i, [_vp, _u64, _u32, _u32]),
    ("mc_index_add_device", _i, [_vp, _vp, _vp, _i, _vp, _u64, _u64, _u64]),
    ("mc_index_add_host", _i, [_vp, _vp, _vp, _i, _vp, _u64, _u64, _u64]),
    ("mc_index_next_pass", _i, [_vp]),
    ("mc_index_end", _i, [_vp]),
    ("mc_index_plan", _i, [_u64, _u32, _u64, C.POINTER(McIndexPlan)]),
    ("mc_get_db_info", _i, [_vp, C.POINTER(McDbInfo)]),
    ("mc_get_stats", _i, [_vp, C.POINTER(McStats)]),
    ("mc_alloc_batches", _i, [_vp, _u32, _u64, _u64, _i]),
    ("mc_batch_buffers", _i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    ("mc_submit", _i, [_vp, _u32, _u64, _u64, _u32]),
    ("mc_wait", _i, [_vp, _u32]),
    ("mc_sync", _i, [_vp]),
    ("mc_free_batches", _i, [_vp]),
    ("mc_query_device", _i, [_vp, _vp, _vp, _u64, _u64, _u32, _vp, _vp, _vp]),
    ("mc_merge_rows_device", _i, [_vp, _vp, _vp, _u64, _vp, _vp]),
    ("mc_result_rows_device", _i, [_vp, _vp, _u64, _vp, _vp]),
    ("mc_merge_result_device", _i, [_vp, C.POINTER(_vp), _u32, _u64, _vp, _vp, _vp]),
    ("mc_text_alloc", _i, [_vp, _u32, _u64, _u64, _u64]),
    ("mc_text_buffers", _i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    ("mc_text_submit", _i, [_vp, _u32, _vp, _u64]),
    ("mc_text_wait", _i, [_vp, _u32, C.POINTER(_u64), C.POINTER(_u32)]),
    ("mc_text_free", _i, [_vp]),
]


def load_library():
    """Load libmcclark.so and bind every symbol of include/mc_api.h."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build it with `make -C jn_cuclark_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % path)
        # One HIP runtime per process.  libmcclark.so needs "libamdhip64.so.7" (found in /opt/rocm through its RUNPATH); torch
        # ships a copy of its own and asks for it by another name, so the loader shares torch's copy with us when torch came
        # first, and loads BOTH when we came first -- and the runtime that starts second finds no device (seen on the GPU box:
        # library loaded for mc_index_plan, then torch.cuda.is_available() True, then mc_open: "no HIP device visible";
        # tools/probe/load_order.py).  This package uses torch for device memory and queues anyway: it goes first.  A program
        # that binds the C ABI without torch (bin/cuCLARK, the reference's main.cc) has one runtime by construction.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)          # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def index_plan(n_keys_total, n_parts, hbm_bytes):
    """mc_index_plan: the loader's arithmetic (fill, lines and bytes per part, smallest part count); no device needed"""
    p = McIndexPlan()
    check(load_library().mc_index_plan(int(n_keys_total), int(n_parts), int(hbm_bytes), C.byref(p)))
    return {f: getattr(p, f) for f, _ in McIndexPlan._fields_}


def check(rc):
    if rc != 0:
        msg = load_library().mc_last_error()
        raise McError(rc, msg.decode() if msg else "")
