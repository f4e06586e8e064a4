"""ctypes binding of libgsmcal.so (include/gsmcal.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)
c_u8_p = C.POINTER(C.c_uint8)

MAX_HITS = 24
MAX_POS_ROWS = 6 * MAX_HITS
TABLE_COLS = 10

# every extern "C" symbol include/gsmcal.h declares: (restype, argtypes)
SIGNATURES = {
    "gsmcal_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gsmcal_ctx_create_on_stream": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gsmcal_ctx_destroy": (None, [C.c_void_p]),
    "gsmcal_sync": (C.c_int, [C.c_void_p]),
    "gsmcal_last_error": (C.c_char_p, [C.c_void_p]),
    "gsmcal_last_call_report": (C.c_long, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "gsmcal_num2str": (C.c_long, [c_double_p, C.c_int, C.c_char_p, C.c_size_t]),
    "gsmcal_version": (C.c_char_p, []),
    "gsmcal_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gsmcal_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsmcal_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gsmcal_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gsmcal_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_profile_reset": (C.c_int, [C.c_void_p]),
    "gsmcal_profile_filter": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gsmcal_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_double_p, c_long_p]),
    "gsmcal_raw2iq": (C.c_int, [C.c_void_p, c_double_p, C.c_long, C.c_int, c_double_p]),
    "gsmcal_raw2iq_u8": (C.c_int, [C.c_void_p, c_u8_p, C.c_long, C.c_int, c_double_p]),
    "gsmcal_chn_filter_8x_4x": (C.c_int, [C.c_void_p, c_double_p, C.c_long, C.c_int, c_double_p, C.c_int, c_double_p]),
    "gsmcal_chn_filter_4x": (C.c_int, [C.c_void_p, c_double_p, C.c_long, C.c_int, c_double_p, C.c_int, c_double_p]),
    "gsmcal_filter": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p, C.c_long, C.c_int, C.c_int, c_double_p]),
    "gsmcal_move_fft_snr_runtime_avg": (C.c_int, [C.c_void_p, c_double_p, C.c_long, C.c_int, C.c_int, C.c_double,
                                                  c_int_p, c_double_p, c_double_p, c_double_p]),
    "gsmcal_specific_fft_snr_fix_avg": (C.c_int, [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_int, C.c_double,
                                                  C.c_double, c_int_p, c_double_p, c_double_p]),
    "gsmcal_FCCH_coarse_position": (C.c_int, [C.c_void_p, c_double_p, C.c_long, C.c_int, c_double_p, c_double_p,
                                              C.c_int, c_int_p]),
    "gsmcal_FCCH_fine_correction": (C.c_int, [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_int, C.c_int,
                                              C.c_double, c_double_p, C.c_int, c_int_p, c_double_p, C.c_long,
                                              c_long_p, c_double_p, c_double_p]),
    "gsmcal_SCH_corr_rate_correction": (C.c_int, [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_int, c_double_p,
                                                  C.c_int, C.c_int, c_double_p, C.c_int, c_int_p, c_double_p,
                                                  C.c_long, c_long_p, c_double_p]),
    "gsmcal_carrier_correct_post_SCH": (C.c_int, [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_int, C.c_int,
                                                  C.c_int, C.c_double, c_double_p, C.c_long, c_long_p, c_double_p]),
    "gsmcal_SCH_equalise": (C.c_int, [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_int, C.c_int, c_double_p, C.c_int,
                                      C.c_int, c_double_p, C.c_int, c_int_p, c_int_p]),
    "gsmcal_total_ppm_calculation": (C.c_int, [c_double_p, C.c_int, c_double_p]),
    "gsmcal_frontend_batch": (C.c_int, [C.c_void_p, c_u8_p, C.c_int, C.c_long, c_double_p, C.c_int, C.c_int,
                                        c_double_p]),
    "gsmcal_frontend_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, c_double_p, C.c_int, C.c_int,
                                            C.c_void_p]),
    "gsmcal_fcch_scan_batch": (C.c_int, [C.c_void_p, c_u8_p, C.c_int, C.c_long, c_double_p, C.c_int, c_double_p,
                                         c_double_p, c_double_p, c_double_p, c_int_p]),
    "gsmcal_fcch_scan_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, c_double_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsmcal_calibrate_batch": (C.c_int, [C.c_void_p, c_u8_p, C.c_int, C.c_long, c_double_p, C.c_int, c_double_p,
                                         C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_long_p]),
    "gsmcal_calibrate_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, c_double_p, C.c_int,
                                             c_double_p, C.c_int, c_double_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "gsmcal_ctx_set_pipeline_depth": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_ctx_get_pipeline_depth": (C.c_int, [C.c_void_p]),
    "gsmcal_ctx_pipeline_queues": (C.c_int, [C.c_void_p]),
    "gsmcal_fused_tail_reruns": (C.c_longlong, [C.c_void_p]),
    "gsmcal_fused_tail_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "gsmcal_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "gsmcal_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gsmcal_comm_init_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gsmcal_last_batch_snr": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_long, c_long_p, c_long_p]),
    "gsmcal_comm_init_file_nonce": (C.c_int, [C.c_void_p, C.c_char_p, C.c_ulonglong, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gsmcal_comm_default_nonce": (C.c_ulonglong, []),
    "gsmcal_comm_id_file_exchange": (C.c_int, [C.c_char_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_double]),
    "gsmcal_comm_id_file_exchange_aged": (C.c_int, [C.c_char_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_double]),
    "gsmcal_comm_id_file_remove": (C.c_int, [C.c_char_p]),
    "gsmcal_comm_destroy": (None, [C.c_void_p]),
    "gsmcal_allgather_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gsmcal_allgather_table_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "gsmcal_allgather_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_allgather_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_ring_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "gsmcal_ring_destroy": (None, [C.c_void_p]),
    "gsmcal_ring_host": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gsmcal_ring_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "gsmcal_ring_acquire": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gsmcal_ring_release": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_ring_host_ready": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmcal_synth_expand_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                          C.c_ulonglong]),
    "gsmcal_last_batch_details": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                            c_double_p, c_int_p]),
}


class Params(C.Structure):
    """gsmcal_params of include/gsmcal.h (field order and types must match)."""
    _fields_ = [("coarse_th_db", C.c_double), ("coarse_mv_factor", C.c_int), ("coarse_max_offset", C.c_int),
                ("min_hits", C.c_int), ("fine_max_offset", C.c_int), ("fine_max_ppm", C.c_double),
                ("fine_gate_snr_db", C.c_double), ("fine_noise_bw_hz", C.c_double), ("sch_max_offset", C.c_int),
                ("sch_max_ppm", C.c_double), ("post_min_bcch", C.c_int), ("scan_min_hits", C.c_int),
                ("scan_spacing", C.c_double), ("scan_spacing_idle", C.c_double), ("scan_tol", C.c_double)]


SIGNATURES.update({
    "gsmcal_params_default": (None, [C.POINTER(Params)]),
    "gsmcal_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "gsmcal_get_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
})


class GsmcalError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def _share_hip_runtime():
    """Two HIP runtimes cannot coexist in one process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so; libgsmcal.so
    names libamdhip64.so.7 and, loaded first, would pull in the system one (/opt/rocm) -- and a later `import torch` would
    then find "No HIP GPUs".  Loaded after torch it simply resolves to torch's copy (what bench.py and most tests do).  To
    make the order irrelevant, the wheel's runtime -- when such a wheel is installed and nothing has loaded a HIP runtime
    yet -- is mapped first, WITHOUT importing torch.  GSMCAL_HIP_RUNTIME=system keeps the system runtime."""
    import sys
    if os.environ.get("GSMCAL_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return                                   # a runtime is already mapped: whichever it is, it is the one
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 - best effort: without it the process just must import torch before gsmcal
        pass


def load(build_if_missing=True):
    """Load libgsmcal.so and attach prototypes.  Raises GsmcalError if it is absent and cannot be built:
    there is no CPU fallback in this package."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise GsmcalError(f"{path} is missing (build it with __graft_entry__.build()); no CPU fallback exists")
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise GsmcalError(f"libgsmcal.so is missing and could not be built: {e}") from e
    _share_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
