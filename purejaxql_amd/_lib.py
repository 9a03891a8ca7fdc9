"""ctypes binding of libpqn_hip.so (the C ABI declared in include/pqn_hotpath.h).

There is NO fallback: if the HIP library is missing the import of any compute
entry point raises.  The oracle under oracle/ is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpqn_hip.so")

c_void_p, c_int, c_int32, c_int64, c_uint32, c_uint64, c_float = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float)


class EnvSpec(C.Structure):
    """pqn_env_spec_t"""
    _fields_ = [("obs_dim", c_int32 * 3), ("obs_size", c_int32), ("num_actions", c_int32),
                ("max_steps", c_int32), ("state_words", c_int32), ("obs_words", c_int32),
                ("canon_si", c_int32), ("canon_sf", c_int32)]


class StepOut(C.Structure):
    """pqn_step_out_t"""
    _fields_ = [("obs", c_void_p), ("obs_bits", c_void_p), ("reward", c_void_p), ("done", c_void_p),
                ("discount", c_void_p), ("returned_episode_returns", c_void_p),
                ("returned_episode_lengths", c_void_p), ("timestep", c_void_p), ("achievements", c_void_p)]


# name -> (restype, argtypes).  Every symbol include/pqn_hotpath.h declares.
SIGNATURES = {
    "pqn_last_error": (C.c_char_p, []),
    "pqn_version": (c_int, []),
    "pqn_threefry2x32": (None, [C.POINTER(c_uint32), C.POINTER(c_uint32), C.POINTER(c_uint32)]),
    "pqn_fold_in": (c_uint64, [c_uint64, c_uint32]),
    "pqn_fold_in_range": (c_int, [c_uint64, c_uint32, c_int32, c_void_p, c_void_p]),
    "pqn_env_id": (c_int, [C.c_char_p]),
    "pqn_env_spec": (c_int, [c_int, C.POINTER(EnvSpec)]),
    "pqn_env_reset": (c_int, [c_int, c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_env_step": (c_int, [c_int, c_int32, c_uint64, c_void_p, c_void_p, c_void_p, C.POINTER(StepOut), c_void_p]),
    "pqn_env_step_optimistic": (c_int, [c_int, c_int32, c_uint64, c_int32, c_void_p, c_void_p, c_void_p, C.POINTER(StepOut),
                                        c_void_p, c_void_p, c_void_p]),
    "pqn_env_export_state": (c_int, [c_int, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_env_import_state": (c_int, [c_int, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_eps_greedy": (c_int, [c_void_p, c_int32, c_int32, c_float, c_uint64, c_void_p, c_void_p, c_void_p]),
    "pqn_q_lambda": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int32, c_int32, c_int32,
                             c_void_p, c_void_p]),
    "pqn_shuffle_keys": (c_int, [c_uint64, c_int32, c_void_p, c_void_p]),
    "pqn_radam_clip_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float,
                                    C.c_double, c_float, c_void_p, c_void_p, c_void_p]),
    "pqn_cnn_layout": (c_int, [c_int32, c_int32, c_void_p]),
    "pqn_cnn_layout_ex": (c_int, [c_int32, c_int32, c_int32, c_void_p]),
    "pqn_qnet_cnn_forward": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                     c_uint64, c_void_p]),
    "pqn_qnet_cnn_workspace_floats": (c_int64, [c_void_p, c_int32]),
    "pqn_qnet_cnn_grad": (c_int, [c_void_p, c_int32] + [c_void_p] * 11 + [c_void_p]),
    "pqn_qnet_cnn_grad_seeds": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_void_p, c_void_p, c_void_p]),
    "pqn_qnet_cnn_apply": (c_int, [c_void_p] * 7 + [c_float, c_float, C.c_double, c_float] + [c_void_p, c_void_p, c_int32, c_void_p]),
    "pqn_qnet_cnn_pack_w1b": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_prof_enable": (c_int, [c_int32]),
    "pqn_prof_read": (c_int, [c_void_p, c_void_p]),
    "pqn_update_sort_temp_bytes": (c_int64, [c_int32]),
    "pqn_cnn_update_workspace_floats": (c_int64, [c_void_p, c_int32, c_int32, c_int32]),
    "pqn_cnn_update": (c_int, [c_void_p, c_void_p]),
    "pqn_cnn_update_phase": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "pqn_mlp_update": (c_int, [c_void_p, c_void_p]),
    "pqn_bigmlp_update": (c_int, [c_void_p, c_void_p]),
    "pqn_peer_region_bytes": (c_int64, [c_int64]),
    "pqn_peer_alloc": (c_int, [c_int64, c_void_p, c_void_p]),
    "pqn_peer_open": (c_int, [c_void_p, c_void_p]),
    "pqn_peer_close": (c_int, [c_void_p]),
    "pqn_peer_free": (c_int, [c_void_p]),
    "pqn_peer_allreduce_mean": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pqn_peer_status": (c_int, [c_void_p, c_void_p]),
    "pqn_mlp_update_seeds": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "pqn_cnn_rollout_seeds": (c_int, [c_int, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float,
                                      c_void_p]),
    "pqn_cnn_update_seeds": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "pqn_cnn_update_seed_groups": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_stream_create_masked": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "pqn_stream_destroy": (c_int, [c_void_p]),
    "pqn_cnn_rollout": (c_int, [c_int, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "pqn_debug_t1_stamps": (c_int, [c_void_p]),
    "pqn_debug_t2_stamps": (c_int, [c_void_p]),
    "pqn_debug_pos_stamps": (c_int, [c_void_p]),
    "pqn_cnn_seed_group": (c_int, [c_int, c_int]),
    "pqn_set_option": (c_int, [C.c_char_p, c_int32]),
    "pqn_get_option": (c_int, [C.c_char_p, c_void_p]),
    "pqn_options_epoch": (c_int, []),
    "pqn_cnn_last_kernel_form": (c_int, [c_void_p, c_void_p]),
    "pqn_mlp_layout": (c_int, [c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pqn_mlp_forward": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64,
                                c_void_p]),
    "pqn_mlp_workspace_floats": (c_int64, [c_void_p, c_int32]),
    "pqn_mlp_grad": (c_int, [c_void_p, c_int32] + [c_void_p] * 11 + [c_void_p]),
    "pqn_mlp_apply": (c_int, [c_void_p] * 7 + [c_float, c_float, C.c_double, c_float] + [c_void_p, c_void_p, c_int32, c_void_p]),
    "pqn_mlp_refresh_transposed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_bigmlp_layout": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pqn_bigmlp_workspace_floats": (c_int64, [c_void_p, c_int32, c_int32]),
    "pqn_bigmlp_weight_plane_floats": (c_int64, [c_void_p]),
    "pqn_bigmlp_refresh_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_bigmlp_refresh_planes_streams": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pqn_bigmlp_forward": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p, c_void_p]),
    "pqn_bigmlp_grad": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "pqn_bigmlp_workspace_view": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "pqn_debug_bm_stamps": (c_int, [c_void_p]),
    "pqn_bigmlp_gemm_scratch_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "pqn_bigmlp_gemm": (c_int, [c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p,
                                c_void_p, c_int64, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load libpqn_hip.so or raise loudly.  No CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C purejaxql_amd/csrc`).  purejaxql_amd has no CPU fallback.")
    # torch bundles its own libamdhip64.so (soname libamdhip64.so.7).  Import torch FIRST so
    # our NEEDED libamdhip64.so.7 binds to that already-loaded runtime; loading /opt/rocm's copy
    # first would put two HIP runtimes in one process (kernels then see "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pqn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'libpqn_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Raw device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    """hipStream_t of torch's current stream, as an integer for void*."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def fold_in(key: int, data: int) -> int:
    return int(load().pqn_fold_in(c_uint64(key & 0xFFFFFFFFFFFFFFFF), c_uint32(data & 0xFFFFFFFF)))


def prng_key(seed: int) -> int:
    """PRNGKey(seed): key words (seed >> 32, seed & 0xffffffff) packed in a uint64."""
    return int(seed) & 0xFFFFFFFFFFFFFFFF


KERNEL_FORMS = {0: "none", 1: "single", 2: "pair", 5: "ksplit", 6: "pos"}


def set_option(name: str, value: int):
    """Run-time switch of the kernel selection (include/pqn_hotpath.h: pqn_set_option)."""
    check(load().pqn_set_option(name.encode(), int(value)), "pqn_set_option")


def get_option(name: str) -> int:
    v = c_int32(0)
    check(load().pqn_get_option(name.encode(), C.addressof(v)), "pqn_get_option")
    return int(v.value)


def last_kernel_form():
    """(training, rollout) kernel forms of the last enqueued launches, as names from KERNEL_FORMS."""
    a, b = c_int32(0), c_int32(0)
    check(load().pqn_cnn_last_kernel_form(C.addressof(a), C.addressof(b)), "pqn_cnn_last_kernel_form")
    return KERNEL_FORMS.get(a.value, str(a.value)), KERNEL_FORMS.get(b.value, str(b.value))


class options:
    """Context manager: with _lib.options(t1_pair=2, bwd_pos=0): ... -- restores the previous values on exit."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False
