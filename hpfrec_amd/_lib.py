"""ctypes binding of libhpf_hip.so (C ABI declared in include/hpf_hip.h).

The library is the product: there is no CPU fallback.  If it is missing or cannot be
loaded, every compute entry point raises -- loudly -- instead of degrading.

torch is imported first on purpose: the PyTorch ROCm wheel bundles its own
libamdhip64.so (SONAME libamdhip64.so.7); loading it before libhpf_hip.so makes the
dynamic linker resolve our NEEDED libamdhip64.so.7 to that same runtime, so streams and
device pointers are shared between torch and the kernels.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
# HPF_HIP_SO: load a differently-tuned build of the same source (kernel A/B runs); default in-tree build
SO_PATH = os.environ.get("HPF_HIP_SO") or os.path.join(_PKG, "libhpf_hip.so")
SRC_PATH = os.path.join(_PKG, "csrc", "hpf_hip.hip")          # the kernels + their launchers
SHARD_SRC_PATH = os.path.join(_PKG, "csrc", "hpf_shard.hip")  # host code: one rank's sharded iteration, RCCL binding
MT_SRC_PATH = os.path.join(_PKG, "csrc", "hpf_mt19937.hip")   # the MT19937 stream of the initial draws (jump-ahead)
SVI_SRC_PATH = os.path.join(_PKG, "csrc", "hpf_svi_prep.hip") # index structures of a stochastic batch, on the device
P2P_SRC_PATH = os.path.join(_PKG, "csrc", "hpf_p2p.hip")      # peer-mapped exchange regions (hipIpc*) + their primitives
SOURCES = (SRC_PATH, SHARD_SRC_PATH, MT_SRC_PATH, SVI_SRC_PATH, P2P_SRC_PATH)
HEADERS = (os.path.join(_PKG, "csrc", "hpf_p2p_dev.h"),)
INC_PATH = os.path.join(_ROOT, "include")

HPF_HIP_ABI_VERSION = 24

#: every symbol include/hpf_hip.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "hpf_hip_abi_version", "hpf_hip_ld_for_k", "hpf_hip_device_info", "hpf_hip_sweep_f32",
    "hpf_hip_sweep_finalize_f32",
    "hpf_hip_row_finalize_f32", "hpf_hip_row_finalize_ranges_f32", "hpf_hip_colsum_reduce_f32", "hpf_hip_colsum_f32", "hpf_hip_colsum_sequential_f32", "hpf_hip_expect_f32",
    "hpf_hip_segsum_f32", "hpf_hip_pair_llk_f32", "hpf_hip_llk_sweep_f32", "hpf_hip_pair_dot_f32", "hpf_hip_score_rows_f32", "hpf_hip_gather_probe_f32",
    "hpf_hip_svi_shape_rows_f32", "hpf_hip_svi_refresh_f32", "hpf_hip_svi_rate_rows_f32", "hpf_hip_svi_side_f32", "hpf_hip_sweep_svi_f32", "hpf_hip_sweep_svi_batch_f32", "hpf_hip_mt19937_words", "hpf_hip_uniform_rows_f32", "hpf_hip_svi_batch_prepare", "hpf_hip_svi_prep_scratch_words", "hpf_hip_svi_batch_sizeof", "hpf_hip_svi_epoch_prepare", "hpf_hip_svi_epoch_scratch_words", "hpf_hip_svi_epoch_sizeof", "hpf_hip_svi_coo_sizeof", "hpf_hip_svi_coo_narrow", "hpf_hip_svi_coo_prepare", "hpf_hip_segsum_desc_f32", "hpf_hip_fold_in_f32",
    "hpf_hip_item_shape_rows_f32", "hpf_hip_item_apply_rows_f32", "hpf_hip_gather_payload_ld", "hpf_hip_rccl_open", "hpf_hip_rccl_unique_id", "hpf_hip_rccl_comm_init", "hpf_hip_rccl_comm_count",
    "hpf_hip_rccl_comm_destroy", "hpf_hip_rccl_all_reduce_f32", "hpf_hip_rccl_reduce_scatter_f32",
    "hpf_hip_rccl_all_gather_f32", "hpf_hip_shard_plan_create", "hpf_hip_shard_plan_destroy", "hpf_hip_shard_iterate",
    "hpf_hip_shard_join", "hpf_hip_shard_status", "hpf_hip_shard_exchange_only", "hpf_hip_shard_desc_layout", "hpf_hip_shard_trace",
    "hpf_hip_mt19937_scratch_words", "hpf_hip_mt19937_jump_poly",
    "hpf_hip_p2p_ctrl_bytes", "hpf_hip_p2p_region_create", "hpf_hip_p2p_region_handles", "hpf_hip_p2p_region_connect",
    "hpf_hip_p2p_region_data", "hpf_hip_p2p_region_set_timeout", "hpf_hip_p2p_region_next_epoch",
    "hpf_hip_p2p_region_status", "hpf_hip_p2p_region_destroy", "hpf_hip_p2p_signal", "hpf_hip_p2p_wait",
    "hpf_hip_p2p_allreduce_vec_f32", "hpf_hip_p2p_pull_f32",
)

_lib = None


class HpfHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if (not force) and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= max(
            max(os.path.getmtime(p) for p in SOURCES + HEADERS), os.path.getmtime(os.path.join(INC_PATH, "hpf_hip.h"))):
        return SO_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I" + INC_PATH,
           "-o", SO_PATH] + list(SOURCES) + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HpfHipError(
            "hpfrec_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    try:
        L = ctypes.CDLL(SO_PATH)
    except OSError as e:  # pragma: no cover
        raise HpfHipError("hpfrec_amd: cannot load %s: %s" % (SO_PATH, e))
    vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    L.hpf_hip_abi_version.argtypes = []
    L.hpf_hip_ld_for_k.argtypes = [ci]
    L.hpf_hip_device_info.argtypes = [ctypes.POINTER(ci), ctypes.c_char_p, ci]
    L.hpf_hip_sweep_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]
    L.hpf_hip_sweep_finalize_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, ci,
                                             ci, ci, vp]
    L.hpf_hip_row_finalize_f32.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, ci, ci, ci,
                                           ci, vp]
    L.hpf_hip_row_finalize_ranges_f32.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, ci,
                                                  ci, ci, ci, ci, vp]
    L.hpf_hip_gather_payload_ld.argtypes = [ci]
    L.hpf_hip_item_shape_rows_f32.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, ci, ci, ci, vp]
    L.hpf_hip_item_apply_rows_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, i64, ci, vp, vp, ci, vp]
    L.hpf_hip_rccl_open.argtypes = [ctypes.c_char_p]
    L.hpf_hip_rccl_unique_id.argtypes = [vp]
    L.hpf_hip_rccl_comm_init.argtypes = [ctypes.POINTER(vp), ci, ci, vp]
    L.hpf_hip_rccl_comm_count.argtypes = [vp, ctypes.POINTER(ci)]
    L.hpf_hip_rccl_comm_destroy.argtypes = [vp]
    L.hpf_hip_rccl_all_reduce_f32.argtypes = [vp, vp, i64, vp]
    L.hpf_hip_rccl_reduce_scatter_f32.argtypes = [vp, vp, vp, i64, vp]
    L.hpf_hip_rccl_all_gather_f32.argtypes = [vp, vp, vp, i64, vp]
    L.hpf_hip_shard_desc_layout.argtypes = [vp]
    L.hpf_hip_shard_plan_create.argtypes = [vp, ctypes.POINTER(vp)]
    L.hpf_hip_shard_plan_destroy.argtypes = [vp]
    L.hpf_hip_shard_iterate.argtypes = [vp, vp, vp, ci, vp]
    L.hpf_hip_shard_join.argtypes = [vp, vp]
    L.hpf_hip_shard_status.argtypes = [vp]
    L.hpf_hip_shard_exchange_only.argtypes = [vp, ci, ci, vp]
    L.hpf_hip_shard_trace.argtypes = [vp, vp, i64, ctypes.POINTER(i64)]
    L.hpf_hip_colsum_reduce_f32.argtypes = [vp, ci, vp, ci, vp]
    L.hpf_hip_colsum_f32.argtypes = [vp, i64, ci, vp, ci, vp]
    L.hpf_hip_expect_f32.argtypes = [vp, vp, vp, vp, vp, i64, ci, ci, vp, vp, cf, vp, vp]
    L.hpf_hip_colsum_sequential_f32.argtypes = [vp, i64, ci, vp, vp]
    L.hpf_hip_segsum_f32.argtypes = [vp, vp, vp, i64, vp, ci, ci, ci, vp]
    L.hpf_hip_pair_llk_f32.argtypes = [vp, vp, vp, vp, vp, i64, vp, ci, ci, ci, ci, vp]
    L.hpf_hip_llk_sweep_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.hpf_hip_pair_dot_f32.argtypes = [vp, vp, vp, vp, i64, vp, ci, ci, vp]
    L.hpf_hip_score_rows_f32.argtypes = [vp, vp, i64, vp, ci, ci, vp]
    L.hpf_hip_mt19937_words.argtypes = [vp, vp, i64, vp, vp]
    L.hpf_hip_mt19937_scratch_words.argtypes = [i64]
    L.hpf_hip_mt19937_jump_poly.argtypes = [ci, vp]
    L.hpf_hip_svi_batch_prepare.argtypes = [vp, vp]
    L.hpf_hip_svi_prep_scratch_words.argtypes = []
    L.hpf_hip_svi_batch_sizeof.argtypes = []
    L.hpf_hip_svi_epoch_prepare.argtypes = [vp, vp]
    L.hpf_hip_svi_epoch_scratch_words.argtypes = [ci]
    L.hpf_hip_svi_epoch_sizeof.argtypes = []
    L.hpf_hip_svi_coo_sizeof.argtypes = []
    L.hpf_hip_svi_coo_narrow.argtypes = [vp, i64, i64, vp, vp, vp]
    L.hpf_hip_svi_coo_prepare.argtypes = [vp, vp]
    L.hpf_hip_segsum_desc_f32.argtypes = [vp, vp, vp, i64, vp, ci, vp]
    L.hpf_hip_fold_in_f32.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, ci, ci, ci, vp]
    L.hpf_hip_uniform_rows_f32.argtypes = [vp, vp, vp, vp, i64, cf, cf, ci, ci, vp]
    L.hpf_hip_gather_probe_f32.argtypes = [vp, i64, vp, vp, ci, vp]
    L.hpf_hip_svi_shape_rows_f32.argtypes = [vp, i64, vp, vp, vp, cf, cf, cf, ci, ci, ci, vp]
    L.hpf_hip_svi_refresh_f32.argtypes = [i64, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, ci, ci, ci, ci, ci, vp]
    L.hpf_hip_svi_side_f32.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, cf, cf, ci, ci, ci, ci,
                                       ci, vp, vp, vp, ci, vp]
    L.hpf_hip_sweep_svi_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, cf, cf,
                                        ci, ci, ci, ci, vp, vp]
    L.hpf_hip_sweep_svi_batch_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp,
                                              cf, cf, cf, cf, cf, cf, cf, ci, ci, ci, ci, vp, vp]
    L.hpf_hip_svi_rate_rows_f32.argtypes = [vp, i64, vp, vp, vp, vp, cf, cf, cf, cf, ci, ci, ci, vp]
    u32 = ctypes.c_uint32
    L.hpf_hip_p2p_ctrl_bytes.argtypes = [ci]
    L.hpf_hip_p2p_region_create.argtypes = [ci, ci, ci, i64, ctypes.POINTER(vp)]
    L.hpf_hip_p2p_region_handles.argtypes = [vp, vp]
    L.hpf_hip_p2p_region_connect.argtypes = [vp, vp]
    L.hpf_hip_p2p_region_data.argtypes = [vp, ci, ctypes.POINTER(vp)]
    L.hpf_hip_p2p_region_set_timeout.argtypes = [vp, cf]
    L.hpf_hip_p2p_region_next_epoch.argtypes = [vp, ctypes.POINTER(u32)]
    L.hpf_hip_p2p_region_status.argtypes = [vp, ctypes.POINTER(u32)]
    L.hpf_hip_p2p_region_destroy.argtypes = [vp]
    L.hpf_hip_p2p_signal.argtypes = [vp, ci, u32, vp]
    L.hpf_hip_p2p_wait.argtypes = [vp, ci, u32, u32, vp]
    L.hpf_hip_p2p_allreduce_vec_f32.argtypes = [vp, ci, u32, vp, vp]
    L.hpf_hip_p2p_pull_f32.argtypes = [vp, ci, u32, ci, i64, vp, i64, ci, vp]
    for s in SYMBOLS:
        getattr(L, s).restype = ci
    L.hpf_hip_p2p_ctrl_bytes.restype = i64
    L.hpf_hip_mt19937_scratch_words.restype = i64
    L.hpf_hip_svi_prep_scratch_words.restype = i64
    L.hpf_hip_svi_batch_sizeof.restype = i64
    L.hpf_hip_svi_epoch_sizeof.restype = i64
    L.hpf_hip_svi_coo_sizeof.restype = i64
    L.hpf_hip_svi_epoch_scratch_words.restype = i64
    if L.hpf_hip_abi_version() != HPF_HIP_ABI_VERSION:
        raise HpfHipError("hpfrec_amd: ABI version mismatch, rebuild libhpf_hip.so")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise HpfHipError("hpfrec_amd: %s failed with code %d" % (what, rc))


def ld_for_k(k):
    """Padded leading dimension of a factor table for k latent factors (pure function of k;
    mirrors hpf_hip_ld_for_k so host-only code paths do not need the .so)."""
    if k <= 0:
        raise ValueError("k must be positive")
    ld = 32
    while ld < k:
        ld <<= 1
    if ld > 1024:
        raise ValueError("k=%d is larger than the kernels are instantiated for (max 1024)" % k)
    return ld


def device_info():
    cu = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(lib().hpf_hip_device_info(ctypes.byref(cu), buf, 256), "hpf_hip_device_info")
    return cu.value, buf.value.decode()
