"""ctypes binding of the C-ABI engine (include/mfas_hip.h, built from mfas_amd/csrc/mfas_hip.hip).

There is deliberately no fallback: if ``libmfas_hip.so`` is missing or fails to load, importing
anything that needs the engine raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFAS_LIB") or os.path.join(_HERE, "csrc", "libmfas_hip.so")   # MFAS_LIB: debug builds

MFAS_DT = {"float32": 0, "bfloat16": 1, "float16": 2}
MAX_TAPS = 8        # MFAS_MAX_TAPS


class mfas_hyper(C.Structure):
    _fields_ = [("R", C.c_int32), ("C", C.c_int32), ("B", C.c_int32), ("bn", C.c_int32),
                ("alphas", C.c_int32), ("multitask", C.c_int32), ("drpt", C.c_double),
                ("wd", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("adam_eps", C.c_double), ("bn_eps", C.c_double), ("bn_momentum", C.c_double),
                ("s_sizes", C.c_int32 * MAX_TAPS), ("v_sizes", C.c_int32 * MAX_TAPS), ("loss_mode", C.c_int32),
                ("allow_plain_cell", C.c_int32), ("f1_threshold", C.c_double), ("tap_bits", C.c_int32), ("order_per_candidate", C.c_int32)]


class mfas_table(C.Structure):
    _fields_ = [("s", C.c_void_p * MAX_TAPS), ("v", C.c_void_p * MAX_TAPS), ("vlogit", C.c_void_p),
                ("slogit", C.c_void_p), ("label", C.c_void_p), ("multilabel", C.c_void_p), ("N", C.c_int64),
                ("dtype", C.c_int32), ("_pad", C.c_int32)]


class mfas_epoch_stats(C.Structure):
    _fields_ = [("train_loss_sum", C.c_double), ("dev_loss_sum", C.c_double),
                ("train_corrects", C.c_int64), ("dev_corrects", C.c_int64)]


_lib = None


def lib():
    """Load the engine once; raise loudly if it is not built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"MFAS HIP engine not built: {LIB_PATH} missing "
                           "(run `python -c 'import __graft_entry__ as g; g.build()'`)")
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    L.mfas_last_error.restype = C.c_char_p
    L.mfas_version.restype = C.c_int
    L.mfas_source_digest.restype = C.c_char_p
    L.mfas_population_set_best_threshold.argtypes = [P, C.c_double]
    L.mfas_population_create.argtypes = [C.POINTER(mfas_hyper), P, P, P, C.c_int32, C.c_int32, P,
                                         C.c_int32, C.POINTER(P)]
    L.mfas_population_destroy.argtypes = [P]
    L.mfas_population_destroy.restype = None
    L.mfas_population_param_count.argtypes = [P, C.c_int32]
    L.mfas_population_param_count.restype = C.c_int64
    L.mfas_population_set_params.argtypes = [P, C.c_int32, P]
    L.mfas_population_get_params.argtypes = [P, C.c_int32, C.c_int32, P]
    L.mfas_population_init.argtypes = [P, P]
    L.mfas_population_train.argtypes = [P, C.POINTER(mfas_table), C.POINTER(mfas_table), P, P, C.c_int32,
                                        C.c_int64, C.c_int32, P, P]
    L.mfas_population_forward.argtypes = [P, C.c_int32, C.POINTER(mfas_table), C.c_int64, C.c_int64, P, P]
    L.mfas_population_forward_train.argtypes = [P, C.c_int32, C.POINTER(mfas_table), C.c_int64, C.c_int32, C.c_int32, P]
    L.mfas_population_backward.argtypes = [P, C.c_int32, C.POINTER(mfas_table), C.c_int64, C.c_int32, C.c_int32, P]
    L.mfas_population_sweep_profile.argtypes = [P, P, P, P]
    L.mfas_population_schedule.argtypes = [P, P]
    L.mfas_population_plan.argtypes = [C.POINTER(mfas_hyper), P, P, C.c_int32, C.c_int32, C.c_int32, P]
    L.mfas_population_set_profiling.argtypes = [P, C.c_int32]
    L.mfas_population_set_pos_weight.argtypes = [P, P]
    L.mfas_stream_probe.argtypes = [C.c_int64, C.c_int32, P]
    L.mfas_global_pool.argtypes = [P, C.c_int32, C.c_int64, C.c_int64, P, C.c_int32, P]
    L.mfas_range_push.argtypes = [C.c_char_p]
    L.mfas_population_init_torch_streams.argtypes = [P, P, P, C.c_double, C.c_double]
    L.mfas_tuning_describe.argtypes = [C.c_char_p, C.c_int32]
    _lib = L
    return L


EXPORTS = ["mfas_last_error", "mfas_version", "mfas_population_create", "mfas_population_destroy",
           "mfas_population_param_count", "mfas_population_set_params", "mfas_population_get_params",
           "mfas_population_init", "mfas_population_train", "mfas_population_forward",
           "mfas_population_sweep_profile", "mfas_population_set_profiling", "mfas_population_set_pos_weight", "mfas_global_pool", "mfas_stream_probe", "mfas_source_digest",
           "mfas_population_set_best_threshold", "mfas_population_forward_train", "mfas_population_schedule",
           "mfas_population_plan", "mfas_population_backward", "mfas_range_push", "mfas_range_pop", "mfas_population_init_torch_streams", "mfas_tuning_describe"]


def tuning() -> dict:
    """The engine's environment switches as the library parses them right now (mfas_tuning_describe): {name: value string}."""
    buf = C.create_string_buffer(2048)
    check(lib().mfas_tuning_describe(buf, 2048))
    return dict(kv.split("=", 1) for kv in buf.value.decode().split())


import contextlib


@contextlib.contextmanager
def profiler_range(name: str):
    """A roctx range around a host-side region (mfas_range_push / mfas_range_pop: no-ops without the marker library)."""
    L = lib()
    L.mfas_range_push(name.encode())
    try:
        yield
    finally:
        L.mfas_range_pop()


def check(rc):
    if rc != 0:
        raise RuntimeError(f"mfas_hip error {rc}: {lib().mfas_last_error().decode()}")
