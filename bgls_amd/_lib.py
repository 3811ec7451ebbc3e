"""ctypes binding of the C ABI (include/bgls_hip.h).  Loads the in-tree HIP library and fails
loudly if it is missing: there is no CPU fallback for the product path."""
import ctypes
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGLS_LIB_VARIANT=legacy loads libbgls_hip_legacy.so (`make LEGACY=1`: the shipped library plus the 32-bit-limb fallback kernels that only
# BGLS_LEGACY / BGLS_MILLER_SHAPE=5 reach) -- tests/test_gpu_legacy_paths.py and A/B runs; nothing else should set it.
_VARIANT = os.environ.get("BGLS_LIB_VARIANT", "")
if _VARIANT not in ("", "legacy"):
    raise RuntimeError("bgls_amd: unknown BGLS_LIB_VARIANT %r" % _VARIANT)
LIB_PATH = os.path.join(_HERE, "libbgls_hip%s.so" % ("_" + _VARIANT if _VARIANT else ""))

u8p = ctypes.POINTER(ctypes.c_uint8)
u64p = ctypes.POINTER(ctypes.c_uint64)
vp = ctypes.c_void_p
sz = ctypes.c_size_t
ci = ctypes.c_int

# name -> (restype, argtypes); mirrors include/bgls_hip.h one to one
SIGNATURES = {
    "bgls_init": (ci, [ci]),
    "bgls_last_error": (ctypes.c_char_p, []),
    "bgls_abi_version": (ci, []),
    "bgls_fp_size": (sz, [ci]),
    "bgls_g1_size": (sz, [ci]),
    "bgls_g2_size": (sz, [ci]),
    "bgls_gt_size": (sz, [ci]),
    "bgls_verify_aggregate": (ci, [ci, u8p, u8p, u8p, u64p, sz, ci]),
    "bgls_verify_multi": (ci, [ci, u8p, u8p, sz, u8p, sz]),
    "bgls_verify_multi_batch": (ci, [ci, u8p, u8p, u64p, sz, u8p, u64p, ci]),
    "bgls_aggregate_sets": (ci, [ci, ci, u8p, u64p, sz, u8p]),
    "bgls_verify_multi_batch_dev": (ci, [ci, vp, vp, vp, sz, sz, vp, sz, sz, ci, vp]),
    "bgls_verify_multi_batch_submit_dev": (ci, [ci, vp, vp, vp, sz, sz, vp, sz, sz, ci, vp]),
    "bgls_pairing_product": (ci, [ci, u8p, u8p, sz, u8p]),
    "bgls_scale_generator": (ci, [ci, ci, u8p, sz, u8p]),
    "bgls_sign_batch": (ci, [ci, u8p, u8p, u64p, sz, u8p]),
    "bgls_compress_points": (ci, [ci, ci, u8p, sz, u8p]),
    "bgls_decompress_points": (ci, [ci, ci, u8p, sz, u8p, u8p]),
    "bgls_hae_exponents": (ci, [ci, u8p, sz, u8p]),
    "bgls_aggregate_signatures_hae": (ci, [ci, u8p, u8p, sz, u8p]),
    "bgls_verify_multi_hae": (ci, [ci, u8p, u8p, sz, u8p, sz]),
    "bgls_verify_aggregate_hae": (ci, [ci, u8p, u8p, u8p, u64p, sz]),
    "bgls_verify_multi_multiplicity": (ci, [ci, u8p, u8p, ctypes.POINTER(ctypes.c_int64), sz, u8p, sz]),
    "bgls_hash_to_g1": (ci, [ci, u8p, u64p, sz, u8p]),
    "bgls_aggregate_points": (ci, [ci, ci, u8p, sz, u8p]),
    "bgls_scale_points": (ci, [ci, ci, u8p, u8p, u8p, sz, u8p]),
    "bgls_point_add": (ci, [ci, ci, u8p, u8p, u8p]),
    "bgls_point_check": (ci, [ci, ci, u8p]),
    "bgls_check_points": (ci, [ci, ci, u8p, sz, u8p]),
    "bgls_select_device": (ci, [ci]),
    "bgls_keys_upload": (ci, [ci, u8p, sz, ctypes.POINTER(ctypes.c_int), ci, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64)]),
    "bgls_keys_free": (ci, [ctypes.c_uint64]),
    "bgls_keys_info": (ci, [ctypes.c_uint64, ctypes.POINTER(ci), ctypes.POINTER(sz), ctypes.POINTER(ci)]),
    "bgls_verify_aggregate_h": (ci, [ctypes.c_uint64, u8p, u8p, u64p, sz, ci]),
    "bgls_verify_multi_h": (ci, [ctypes.c_uint64, u8p, u8p, sz]),
    "bgls_verify_aggregate_multi": (ci, [ci, u8p, u8p, u8p, u64p, sz, ci, ctypes.POINTER(ctypes.c_int), ci]),
    "bgls_verify_multi_multi": (ci, [ci, u8p, u8p, sz, u8p, sz, ctypes.POINTER(ctypes.c_int), ci]),
    "bgls_last_exchange": (ci, []),
    "bgls_miller_product_keys_dev": (ci, [ctypes.c_uint64, vp, vp, sz, sz, sz, ci, vp, vp, vp]),
    "bgls_verify_multi_keys_dev": (ci, [ctypes.c_uint64, vp, vp, sz, vp]),
    "bgls_verify_multi_keys_submit_dev": (ci, [ctypes.c_uint64, vp, vp, sz, vp]),
    "bgls_rccl_available": (ci, []),
    "bgls_verify_aggregate_h_gt": (ci, [ctypes.c_uint64, u8p, u8p, u64p, sz, ci, u8p]),
    "bgls_generator": (ci, [ci, ci, u8p]),
    "bgls_pair": (ci, [ci, u8p, u8p, u8p]),
    "bgls_gt_mul": (ci, [ci, u8p, u8p, u8p]),
    "bgls_gt_pow": (ci, [ci, u8p, u8p, ci, u8p]),
    "bgls_gt_identity": (ci, [ci, u8p]),
    "bgls_miller_product_dev": (ci, [ci, vp, vp, vp, sz, sz, sz, ci, vp, vp, vp]),
    "bgls_duplicate_scan_dev": (ci, [vp, sz, sz, sz, vp, vp]),
    "bgls_duplicate_scan_bucket_dev": (ci, [vp, sz, sz, sz, ctypes.c_uint, ctypes.c_uint, vp, vp]),
    "bgls_message_digests_dev": (ci, [vp, sz, sz, sz, vp, vp]),
    "bgls_digest_pack_dev": (ci, [vp, sz, ctypes.c_uint, sz, vp, vp, vp]),
    "bgls_duplicate_scan_packed_dev": (ci, [vp, sz, ctypes.c_uint, ctypes.c_uint, vp, vp]),
    "bgls_final_verify_submit_dev": (ci, [ci, vp, sz, vp, vp]),
    "bgls_final_verify_collect": (ci, [ci]),
    "bgls_select_context": (ci, [ci]),
    "bgls_set_throughput_mode": (ci, [ci]),
    "bgls_set_miller_shape": (ci, [ci, ci]),
    "bgls_weighted_sum_dev": (ci, [ci, ci, vp, vp, sz, vp, vp]),
    "bgls_set_msm_min": (ci, [sz]),
    "bgls_final_verify_dev": (ci, [ci, vp, sz, vp, vp]),
    "bgls_aggregate_points_dev": (ci, [ci, ci, vp, sz, vp, vp]),
    "bgls_verify_multi_dev": (ci, [ci, vp, vp, sz, vp, sz, vp]),
    "bgls_verify_multi_submit_dev": (ci, [ci, vp, vp, sz, vp, sz, vp]),
    "bgls_profile_enable": (ci, [ci]),
    "bgls_profile_get": (ci, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]),
    "bgls_probe_mad_peak": (ci, [ctypes.POINTER(ctypes.c_double)]),
    "bgls_selftest_exception_barrier": (ci, [ci]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "bgls_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The HIP library is the product; there is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7.  If this library pulled in
        # /opt/rocm's copy first and torch initialised later, torch would report "No HIP GPUs are available".  Importing
        # torch first lets the loader resolve our DT_NEEDED libamdhip64.so.7 to the copy already mapped (same SONAME).
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def buf(data):
    """bytes-like -> ctypes uint8 array (copy)."""
    b = bytes(data)
    return (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * max(n, 1))()


def last_error():
    return load().bgls_last_error().decode()
