"""ctypes loader for libgmrm_hip.so (C ABI: include/gmrm_hip.h)."""
import ctypes as C
import os
import sys
import warnings
from pathlib import Path

HERE = Path(__file__).resolve().parent
_LIB = None


class GmrmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgmrm_hip error {code}: {msg}")
        self.code = code


c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_u8_p = C.POINTER(C.c_uint8)

KMAX = 8
GMAX = 64


class SweepIn(C.Structure):
    _fields_ = [("G", C.c_int), ("K", C.c_int), ("order", c_int_p), ("sigmag", c_double_p),
                ("pi_est", c_double_p), ("cva", c_double_p), ("sigmae", C.c_double),
                ("rng_state", C.c_uint32 * 624), ("rng_index", C.c_int), ("first", C.c_int), ("count", C.c_int)]


class SweepOut(C.Structure):
    _fields_ = [("cass", c_int_p), ("rng_state", C.c_uint32 * 624), ("rng_index", C.c_int),
                ("n_updates", C.c_longlong), ("n_batches", C.c_longlong), ("device_ms", C.c_double),
                ("n_planned_stops", C.c_longlong), ("n_stale_dots", C.c_longlong), ("n_fast_batches", C.c_longlong),
                ("n_crossed_stops", C.c_longlong), ("n_screen_tries", C.c_longlong), ("n_screened_passes", C.c_longlong)]


class SamplerOpts(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("rank", C.c_int), ("nranks", C.c_int), ("shuffle", C.c_int),
                ("mimic_hydra", C.c_int), ("G", C.c_int), ("K", C.c_int), ("cva", c_double_p),
                ("group_index", c_int_p)]


class HyperC(C.Structure):
    _fields_ = [("sigmae", C.c_double), ("mu", C.c_double), ("m0_sum", C.c_int),
                ("sigmag", C.c_double * GMAX), ("pi_est", C.c_double * (GMAX * KMAX)),
                ("n_updates", C.c_longlong), ("n_batches", C.c_longlong), ("sweep_device_ms", C.c_double),
                ("n_planned_stops", C.c_longlong), ("n_stale_dots", C.c_longlong), ("n_fast_batches", C.c_longlong),
                ("n_crossed_stops", C.c_longlong), ("n_screen_tries", C.c_longlong), ("n_screened_passes", C.c_longlong)]


class GeometryC(C.Structure):
    _fields_ = [("R", C.c_int), ("W", C.c_int), ("conc", C.c_int), ("num_cu", C.c_int), ("max_resident_wg", C.c_int),
                ("hw_queues", C.c_int)]


class IngestStatsC(C.Structure):
    _fields_ = [("bytes", C.c_size_t), ("seconds", C.c_double), ("read_seconds", C.c_double),
                ("threads", C.c_int), ("chunk_bytes", C.c_size_t)]


VP = C.c_void_p
# name -> (restype, argtypes); every symbol include/gmrm_hip.h declares
SIGNATURES = {
    "gmrm_last_error": (C.c_char_p, []),
    "gmrm_abi_version": (C.c_int, []),
    "gmrm_device_count": (C.c_int, []),
    "gmrm_ctx_create": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gmrm_ctx_destroy": (C.c_int, [VP]),
    "gmrm_ctx_sync": (C.c_int, [VP]),
    "gmrm_ctx_geometry": (C.c_int, [VP, C.POINTER(GeometryC)]),
    "gmrm_upload_bed": (C.c_int, [VP, c_u8_p, C.c_size_t, C.c_size_t]),
    "gmrm_load_bed_file": (C.c_int, [VP, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(IngestStatsC)]),
    "gmrm_download_bed": (C.c_int, [VP, c_u8_p, C.c_size_t, C.c_size_t]),
    "gmrm_group_create": (C.c_int, [C.POINTER(VP), C.c_int, C.POINTER(VP), C.POINTER(VP), C.c_int, C.c_int, C.c_int]),
    "gmrm_group_uses_rccl": (C.c_int, [VP]),
    "gmrm_group_iterate": (C.c_int, [VP, C.c_int]),
    "gmrm_group_iterate_steps": (C.c_int, [VP, C.c_int]),
    "gmrm_group_iterate_parts": (C.c_int, [VP, C.c_int, C.c_int]),
    "gmrm_group_destroy": (C.c_int, [VP]),
    "gmrm_rccl_selftest": (C.c_int, [C.c_int]),
    "gmrm_predict_g": (C.c_int, [VP, C.c_int, c_double_p, c_double_p]),
    "gmrm_assoc": (C.c_int, [VP, C.c_int, c_double_p, c_double_p, c_double_p]),
    "gmrm_synth_bed": (C.c_int, [VP, C.c_uint64, C.c_double, C.c_double]),
    "gmrm_synth_bed_ld": (C.c_int, [VP, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_double]),
    "gmrm_phen_prepare": (C.c_int, [c_double_p, c_u8_p, C.c_int, c_double_p, c_u8_p, c_int_p]),
    "gmrm_upload_trait": (C.c_int, [VP, C.c_int, c_double_p, c_u8_p, C.c_int]),
    "gmrm_download_eps": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_upload_eps": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_marker_stats": (C.c_int, [VP, C.c_int]),
    "gmrm_get_marker_stats": (C.c_int, [VP, C.c_int, c_double_p, c_double_p]),
    "gmrm_set_marker_stats": (C.c_int, [VP, C.c_int, c_double_p, c_double_p]),
    "gmrm_dot": (C.c_int, [VP, C.c_int, C.c_int, C.c_double, C.c_double, c_double_p]),
    "gmrm_update_eps": (C.c_int, [VP, C.c_int, C.c_int, c_double_p]),
    "gmrm_update_eps_from": (C.c_int, [VP, C.c_int, VP, C.c_int, c_double_p]),
    "gmrm_offset_eps": (C.c_int, [VP, C.c_int, C.c_double]),
    "gmrm_sumsqr": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_eps_sigma": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_set_groups": (C.c_int, [VP, c_int_p]),
    "gmrm_sweep_launch": (C.c_int, [VP, C.c_int, C.POINTER(SweepIn)]),
    "gmrm_sweep_finish": (C.c_int, [VP, C.c_int, C.POINTER(SweepOut)]),
    "gmrm_get_betas": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_get_comp": (C.c_int, [VP, C.c_int, c_int_p]),
    "gmrm_get_acum": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_set_betas": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_set_comp": (C.c_int, [VP, C.c_int, c_int_p]),
    "gmrm_set_acum": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_sampler_save": (C.c_int, [VP, C.c_char_p, C.c_int]),
    "gmrm_sampler_load": (C.c_int, [VP, C.c_char_p, c_int_p]),
    "gmrm_selftest_math": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int]),
    "gmrm_selftest_shuffle": (C.c_int, [C.c_uint32, C.c_int, c_int_p]),
    "gmrm_eps_snapshot": (C.c_int, [VP, C.c_int]),
    "gmrm_eps_delta_export": (C.c_int, [VP, C.c_int, VP]),
    "gmrm_eps_delta_import": (C.c_int, [VP, C.c_int, VP]),
    "gmrm_sampler_create": (C.c_int, [C.POINTER(VP), VP, C.POINTER(SamplerOpts)]),
    "gmrm_sampler_destroy": (C.c_int, [VP]),
    "gmrm_sampler_init": (C.c_int, [VP]),
    "gmrm_sampler_iterate": (C.c_int, [VP, C.c_int]),
    "gmrm_sampler_draw_mu": (C.c_int, [VP, C.c_int, c_double_p]),
    "gmrm_sampler_begin_sweep": (C.c_int, [VP, c_double_p]),
    "gmrm_sampler_launch_sweep": (C.c_int, [VP, c_double_p]),
    "gmrm_sampler_preshuffle": (C.c_int, [VP]),
    "gmrm_sampler_begin_parts": (C.c_int, [VP, c_double_p]),
    "gmrm_sampler_launch_part": (C.c_int, [VP, C.c_int, C.c_int]),
    "gmrm_sampler_finish_part": (C.c_int, [VP]),
    "gmrm_sampler_end_sweep": (C.c_int, [VP, c_int_p, c_double_p]),
    "gmrm_sampler_epilogue": (C.c_int, [VP, c_int_p, c_double_p]),
    "gmrm_sampler_begin_steps": (C.c_int, [VP, c_double_p]),
    "gmrm_sampler_step": (C.c_int, [VP, C.c_int, c_int_p, c_double_p]),
    "gmrm_sampler_end_steps": (C.c_int, [VP, c_int_p, c_double_p]),
    "gmrm_sampler_abort_steps": (C.c_int, [VP]),
    "gmrm_sampler_adopt": (C.c_int, [VP, C.c_int, c_double_p, c_double_p, C.c_double]),
    "gmrm_sampler_get": (C.c_int, [VP, C.c_int, C.POINTER(HyperC)]),
    "gmrm_sampler_csv_line": (C.c_int, [VP, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
}


def library_path() -> Path:
    return Path(os.environ.get("GMRM_HIP_LIB", HERE / "libgmrm_hip.so"))


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP
    runtimes in one process cannot both own the GPU, and bench.py needs torch.distributed
    (RCCL) beside this library, so when torch is installed its runtime is loaded first and
    libgmrm_hip.so (NEEDED libamdhip64.so.7) binds to it.  GMRM_HIP_RUNTIME=system skips this."""
    if os.environ.get("GMRM_HIP_RUNTIME", "torch") != "torch":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
        if cand.exists():
            return C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except Exception:
        return None
    return None


def load_library():
    """Load libgmrm_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    # eight hardware queues, so that four persistent sweeps on four streams run side by side (capi.cpp, want_hw_queues);
    # set here as well because torch may initialise the HIP runtime before libgmrm_hip.so is loaded
    if "GPU_MAX_HW_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        torch = sys.modules.get("torch")
        try:
            late = torch is not None and torch.cuda.is_initialized()
        except Exception:
            late = False
        if late:                          # the runtime read its environment when torch initialised it: the default (4 queues) holds
            warnings.warn("gmrm_amd was loaded after torch had initialised the HIP runtime: GPU_MAX_HW_QUEUES=8 comes too late, "
                          "more than three phenotype chains will not all sweep side by side; import gmrm_amd (or set the "
                          "variable) before the first torch.cuda call", RuntimeWarning, stacklevel=2)
    _share_torch_hip_runtime()
    if not path.exists():
        raise ImportError(f"{path} is missing: build it with `python -m gmrm_amd.build` "
                          "(hipcc --offload-arch=gfx950). gmrm_amd has no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int) -> int:
    if rc < 0:
        msg = load_library().gmrm_last_error()
        raise GmrmError(rc, msg.decode() if msg else "?")
    return rc
