"""ctypes loader for liblongbow_gpu.so -- the only place the library is opened.

Fails loudly: a missing/unloadable library raises ImportError-like RuntimeError at
first use; a missing GPU raises GPUNotAvailable (the reference's ErrGPUNotAvailable,
internal/gpu/stub.go:10).  Nothing here ever computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LB_GPU_SO selects another build of the library (tools/ use the -DLB_DIAG build, liblongbow_gpu_diag.so)
SO_PATH = os.environ.get("LB_GPU_SO") or os.path.join(_HERE, "liblongbow_gpu.so")

LB_OK = 0
STATUS = {0: "ok", 1: "invalid argument", 2: "index is closed", 3: "GPU not available",
          4: "HIP runtime error", 5: "out of device memory", 6: "unsupported configuration",
          7: "internal error", 8: "context canceled", 9: "context deadline exceeded"}


class LongbowGPUError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"longbow_gpu: {STATUS.get(code, 'error')} (code {code}){': ' + msg if msg else ''}")


class GPUNotAvailable(LongbowGPUError):
    """internal/gpu/stub.go:10 ErrGPUNotAvailable"""


class Canceled(LongbowGPUError):
    """context.Canceled: the call's Cancel fired (internal/store/adaptive_index.go:182 returns ctx.Err())"""


class DeadlineExceeded(LongbowGPUError):
    """context.DeadlineExceeded"""


_lib = None

# every symbol include/longbow_gpu.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
_ip = C.POINTER(C.c_int)
SIGNATURES = [
    ("lb_gpu_device_count", _i, []),
    ("lb_gpu_version", C.c_char_p, []),
    ("lb_gpu_status_string", C.c_char_p, [_i]),
    ("lb_gpu_index_new", _vp, [_i, _i, _i, _ip]),
    ("lb_gpu_index_free", None, [_vp]),
    ("lb_gpu_last_error", C.c_char_p, [_vp]),
    ("lb_gpu_index_set_order", _i, [_vp, _i]),
    ("lb_gpu_index_set_candidate_mode", _i, [_vp, _i]),
    ("lb_gpu_index_set_f16_image", _i, [_vp, _i]),
    ("lb_gpu_index_f16_image_bytes", _i64, [_vp]),
    ("lb_gpu_index_set_search_combining", _i, [_vp, _i]),
    ("lb_gpu_index_combining_stats", _i, [_vp, C.POINTER(C.c_int64)]),
    ("lb_gpu_index_ntotal", _i64, [_vp]),
    ("lb_gpu_index_dim", _i, [_vp]),
    ("lb_gpu_index_device", _i, [_vp]),
    ("lb_cancel_new", _vp, []),
    ("lb_cancel_fire", None, [_vp]),
    ("lb_cancel_set_deadline_ms", None, [_vp, _i64]),
    ("lb_cancel_state", _i, [_vp]),
    ("lb_cancel_free", None, [_vp]),
    ("lb_gpu_index_search_ctx", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp]),
    ("lb_gpu_index_search_device_ctx", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp, _vp]),
    ("lb_simd_distance_batch", _i, [_i, _i, _i, _vp, _i, _vp, _vp, _i64, _vp]),
    ("lb_gpu_pq_search_ctx", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp]),
    ("lb_gpu_pq_search_device_ctx", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp, _vp]),
    ("lb_gpu_pq_set_search_combining", _i, [_vp, _i]),
    ("lb_gpu_pq_combining_stats", _i, [_vp, C.POINTER(C.c_int64)]),
    ("lb_gpu_pq_set_prefilter", _i, [_vp, _i]),
    ("lb_gpu_index_reserve", _i, [_vp, _i64]),
    ("lb_gpu_index_add", _i, [_vp, _i64, _vp, _vp]),
    ("lb_gpu_index_add_device", _i, [_vp, _i64, _vp, _vp]),
    ("lb_gpu_index_search", _i, [_vp, _i64, _vp, _i, _vp, _vp]),
    ("lb_gpu_index_search_device", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp]),
    ("lb_gpu_index_set_filter", _i, [_vp, _vp, _i64]),
    ("lb_gpu_index_filter_int64", _i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _i]),
    ("lb_gpu_index_filter_float32", _i, [_vp, _vp, _i64, C.c_float, _i, _vp, _i64, _i]),
    ("lb_gpu_index_last_fallbacks", _i64, [_vp]),
    ("lb_gpu_index_fused_giveups", _i64, [_vp]),
    ("lb_gpu_index_last_route", _i, [_vp]),
    ("lb_gpu_index_rerank", _i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    ("lb_gpu_index_rerank_device", _i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    ("lb_simd_match_int64", _i, [_i, _vp, _i64, _i64, _i, _vp]),
    ("lb_simd_match_float32", _i, [_i, _vp, _i64, C.c_float, _i, _vp]),
    ("lb_simd_and_bytes", _i, [_i, _vp, _vp, _i64]),
    ("lb_simd_distance_batch_flat", _i, [_i, _i, _i, _vp, _vp, _i64, _i, _vp]),
    ("lb_simd_distance_batch_flat_device", _i, [_i, _i, _i, _vp, _vp, _i64, _i, _vp, _vp]),
    ("lb_gpu_pq_new", _vp, [_i, _vp, _sz, _ip]),
    ("lb_gpu_pq_free", None, [_vp]),
    ("lb_gpu_pq_last_error", C.c_char_p, [_vp]),
    ("lb_gpu_pq_m", _i, [_vp]),
    ("lb_gpu_pq_dims", _i, [_vp]),
    ("lb_gpu_pq_ntotal", _i64, [_vp]),
    ("lb_gpu_pq_add_codes", _i, [_vp, _i64, _vp]),
    ("lb_gpu_pq_add_codes_device", _i, [_vp, _i64, _vp]),
    ("lb_gpu_pq_reserve", _i, [_vp, _i64]),
    ("lb_gpu_pq_get_codes", _i, [_vp, _i64, _i64, _vp]),
    ("lb_gpu_pq_encode", _i, [_vp, _i64, _vp, _vp]),
    ("lb_gpu_pq_encode_device", _i, [_vp, _i64, _vp, _vp, _vp]),
    ("lb_gpu_pq_add_vectors_device", _i, [_vp, _i64, _vp]),
    ("lb_gpu_pq_decode", _i, [_vp, _i64, _vp, _vp]),
    ("lb_gpu_pq_decode_device", _i, [_vp, _i64, _vp, _vp, _vp]),
    ("lb_gpu_pq_rerank", _i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    ("lb_gpu_pq_rerank_device", _i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    ("lb_gpu_pq_set_profiling", _i, [_vp, _i]),
    ("lb_gpu_pq_last_timing", _i, [_vp, _vp]),
    ("lb_gpu_pq_build_adc_table", _i, [_vp, _vp, _vp]),
    ("lb_gpu_pq_adc_distance_batch", _i, [_vp, _vp, _i64, _i64, _vp]),
    ("lb_gpu_pq_search", _i, [_vp, _i64, _vp, _i, _vp, _vp]),
    ("lb_gpu_pq_search_device", _i, [_vp, _i64, _vp, _i, _vp, _vp, _vp]),
    ("lb_gpu_merge_topk_device", _i, [_i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp]),
    ("lb_gpu_merge_topk_packed_device", _i, [_i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    ("lb_gpu_rrf_fuse_device", _i, [_i, _i64, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    ("lb_gpu_rrf_fuse", _i, [_i, _i64, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    ("lb_gpu_fill_uniform_device", _i, [_i, _vp, _i64, _u64, _i64, _vp]),
    ("lb_gpu_fill_codes_device", _i, [_i, _vp, _i64, _u64, _i64, _vp]),
    ("lb_gpu_fill_uniform_rows_device", _i, [_i, _vp, _vp, _i64, _i, _u64, _vp]),
    ("lb_gpu_shader_clock_mhz", C.c_double, [_i, _i]),
    ("lb_flight_datasets_new", _vp, []),
    ("lb_flight_datasets_free", None, [_vp]),
    ("lb_flight_datasets_put", _i, [_vp, C.c_char_p, _vp]),
    ("lb_flight_vector_search_exchange", _i, [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz), C.c_char_p, _sz]),
    ("lb_flight_encode_results", _i, [_vp, _vp, _i64, C.POINTER(_vp), C.POINTER(_sz)]),
    ("lb_flight_free_buffer", None, [_vp]),
    ("lb_flight_index_add_ipc", _i, [_vp, _vp, _sz, C.POINTER(_i64), C.c_char_p, _sz]),
    ("lb_gpu_comm_init_all", _vp, [_i, _vp, _ip]),
    ("lb_gpu_comm_get_unique_id", _i, [_vp]),
    ("lb_gpu_comm_init_rank", _vp, [_i, _i, _i, _vp, _ip]),
    ("lb_gpu_comm_init_host", _vp, [_i, _i, _i, _vp, _vp, _ip]),
    ("lb_gpu_comm_free", None, [_vp]),
    ("lb_gpu_comm_prepare", _i, [_vp, _i64, _i]),
    ("lb_gpu_comm_nranks", _i, [_vp]),
    ("lb_gpu_comm_rank", _i, [_vp]),
    ("lb_gpu_comm_last_error", C.c_char_p, [_vp]),
    ("lb_gpu_comm_search_device", _i, [_vp, _vp, _i64, _vp, _i, _vp, _vp, _vp]),
    ("lb_gpu_comm_search_all", _i, [_vp, _vp, _i64, _vp, _i, _vp, _vp]),
    ("lb_gpu_index_set_profiling", _i, [_vp, _i]),
    ("lb_gpu_index_last_timing", _i, [_vp, _vp, _vp]),
]


def _bind(path, extra=()):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m longbow_amd.build{' --diag' if extra else ''}` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, res, args in list(SIGNATURES) + list(extra):
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """dlopen the HIP library and bind every declared symbol (no GPU needed for this).

    Note for processes that also use PyTorch-ROCm (bench.py, sharded search): import torch BEFORE
    the first call here.  torch ships its own HIP runtime and must initialise first."""
    global _lib
    if _lib is None:
        _lib = _bind(SO_PATH)
    return _lib


# Test hooks and timing-only ablations exist ONLY in the diagnostic build (liblongbow_gpu_diag.so, -DLB_DIAG):
# tests that force a fallback path create their handles on this library (Index(cfg, lib=load_diag())).
DIAG_SO_PATH = os.path.join(_HERE, "liblongbow_gpu_diag.so")
DIAG_SIGNATURES = [
    ("lb_debug_set_sample_tau", None, [_i]),
    ("lb_debug_vmm_fail_next", None, [_i]),
    ("lb_debug_fused_fail_next", None, [_i]),
    ("lb_debug_search_fail_next", None, [_i]),
    ("lb_debug_set_add_register_min", None, [C.c_longlong]),
    ("lb_debug_sample_plan", None, [C.c_longlong, _i, C.c_uint, C.c_uint, C.POINTER(C.c_longlong)]),
]
_diag = None


def load_diag():
    global _diag
    if _diag is None:
        _diag = _bind(DIAG_SO_PATH, DIAG_SIGNATURES)
    return _diag


def require_gpu(device=0):
    lib = load()
    n = lib.lb_gpu_device_count()
    if n <= 0 or device >= n:
        raise GPUNotAvailable(3, f"{n} HIP device(s) visible, device {device} requested")
    return lib


def check(rc, handle=None, pq=False, lib=None):
    if rc == LB_OK:
        return
    msg = ""
    if handle:
        lib = lib or load()
        raw = lib.lb_gpu_pq_last_error(handle) if pq else lib.lb_gpu_last_error(handle)
        msg = raw.decode() if raw else ""
    if rc == 3:
        raise GPUNotAvailable(rc, msg)
    if rc == 8:
        raise Canceled(rc, msg)
    if rc == 9:
        raise DeadlineExceeded(rc, msg)
    raise LongbowGPUError(rc, msg)
