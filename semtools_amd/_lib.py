"""ctypes binding of libsemtools_hip.so (the C ABI in include/semtools_hip.h).

The library is the product; this module only marshals numpy arrays / raw device
pointers across the ABI.  There is NO CPU fallback: if the shared object is
missing or no gfx950 GPU is usable, calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsemtools_hip.so")

SMT_OK, SMT_E_INVALID, SMT_E_HIP, SMT_E_NOMEM, SMT_E_TRUNCATED, SMT_E_IO, SMT_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
MODE_DOCUMENTS, MODE_WORKSPACE = 0, 1
DIM = 256

EXPORTS = [
    "smt_ctx_create", "smt_ctx_create_on_stream", "smt_ctx_aux_stream", "smt_ctx_destroy", "smt_ctx_synchronize", "smt_last_error", "smt_version",
    "smt_device_count", "smt_prof_enable", "smt_prof_reset", "smt_prof_read",
    "smt_model_create", "smt_model_create_from_file", "smt_model_create_from_device", "smt_model_destroy", "smt_embed", "smt_embed_device",
    "smt_corpus_create", "smt_corpus_from_device", "smt_corpus_destroy", "smt_corpus_append_host",
    "smt_corpus_write_rows", "smt_corpus_read_rows", "smt_corpus_truncate", "smt_corpus_rows", "smt_corpus_dim",
    "smt_corpus_prepack", "smt_corpus_image_bytes",
    "smt_corpus_save", "smt_corpus_load", "smt_corpus_append_to_file", "smt_search", "smt_search_topk_device", "smt_merge_topk",
    "smt_merge_topk_device", "smt_merge_topk_packed_device", "smt_ivfpq_build", "smt_ivfpq_destroy", "smt_ivfpq_search", "smt_ivfpq_search_device", "smt_ivfpq_info", "smt_ivfpq_list_sizes", "smt_ivfpq_save", "smt_ivfpq_load", "smt_ivfpq_append",
    "smt_set_tuning", "smt_fnv1a_hash", "smt_line_embedding_id", "smt_doc_meta_id",
    "smt_ctx_uncertain_count", "smt_debug_batched_scores",
    "smt_init", "smt_shutdown", "smt_default_group", "smt_group_create", "smt_group_create_logical", "smt_group_unique_id", "smt_group_create_rank",
    "smt_group_destroy", "smt_group_info", "smt_group_ctx", "smt_group_synchronize", "smt_group_barrier",
    "smt_sharded_corpus_from_host", "smt_sharded_corpus_from_device", "smt_sharded_corpus_load", "smt_sharded_corpus_save",
    "smt_sharded_corpus_destroy", "smt_sharded_corpus_rows", "smt_sharded_corpus_rank_rows", "smt_sharded_corpus_shard",
    "smt_sharded_corpus_append_host", "smt_sharded_search", "smt_sharded_search_topk_device",
    "smt_sharded_ivfpq_build", "smt_sharded_ivfpq_destroy", "smt_sharded_ivfpq_shard", "smt_sharded_ivfpq_search",
    "smt_group_from_ctx", "smt_sharded_corpus_create", "smt_sharded_corpus_load_layout", "smt_sharded_corpus_layout",
    "smt_sharded_corpus_append_to_file", "smt_sharded_corpus_read_rows", "smt_sharded_corpus_write_rows",
    "smt_sharded_model_create", "smt_sharded_model_create_from_file", "smt_sharded_model_destroy", "smt_sharded_embed",
    "smt_sharded_ivfpq_save", "smt_sharded_ivfpq_load", "smt_sharded_ivfpq_append", "smt_sharded_ivfpq_info",
    "smt_group_set_transport", "smt_group_transport", "smt_debug_range_sets", "smt_sharded_corpus_append_to_file_ex",
    "smt_debug_group_fail_next", "smt_search_topk_device_ex", "smt_sharded_search_topk_device_ex", "smt_debug_deliveries",
]
STATUS_PROVED, STATUS_UNCERTAIN, STATUS_OVERFLOW, STATUS_INVALID_QUERY = 0, 1, 2, 3
TRANSPORT_RCCL, TRANSPORT_COPY, TRANSPORT_PEER = 0, 1, 2
TRANSPORT_NAMES = {TRANSPORT_RCCL: "rccl", TRANSPORT_COPY: "copy", TRANSPORT_PEER: "peer"}
UNIQUE_ID_BYTES = 128
HOST_EXPORTS = [
    "smt_host_model_create", "smt_host_model_from_dir", "smt_host_model_destroy", "smt_host_encode",
    "smt_host_search_files", "smt_host_search_content", "smt_host_search_workspace", "smt_host_session_open",
    "smt_host_session_search", "smt_host_session_lines", "smt_host_session_close", "smt_host_workspace_use",
    "smt_host_workspace_status", "smt_host_workspace_prune", "smt_host_workspace_reembed", "smt_host_free", "smt_host_timing_json", "smt_host_tokenizer_load", "smt_host_tokenizer_free",
    "smt_host_tokenizer_encode", "smt_host_tokenizer_info", "smt_host_format_float",
    "smt_host_split_lines", "smt_host_to_lowercase",
    "smt_host_group_from_spec", "smt_host_model_create_group", "smt_host_model_from_dir_group",
    "smt_host_workspace_use_group", "smt_host_workspace_status_group", "smt_host_workspace_prune_group",
]
TOKENIZE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64,
                          C.POINTER(C.c_uint64))


class SmtRange(C.Structure):
    _fields_ = [("begin", C.c_uint64), ("end", C.c_uint64)]


class SmtIvfPqParams(C.Structure):
    _fields_ = [("nlist", C.c_uint32), ("m", C.c_uint32), ("nbits", C.c_uint32), ("train_iters", C.c_uint32),
                ("train_sample", C.c_uint64), ("reserved", C.c_uint32), ("local_pca", C.c_uint32)]


class SmtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsemtools_hip error {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(_HERE, "lib")) if os.path.isdir(os.path.join(_HERE, "lib")) else []:
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(_HERE, "lib", f))
    subprocess.check_call(["bash", script])
    return LIB_PATH


_lib = None


def lib():
    """Load libsemtools_hip.so.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmtError(SMT_E_HIP, f"{LIB_PATH} not built; run semtools_amd/csrc/build.sh (no CPU fallback exists)")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (SONAME libamdhip64.so.7,
    # the same SONAME the system ROCm one has).  If torch is going to be used in this process it must
    # be loaded FIRST so that our NEEDED libamdhip64.so.7 resolves to the copy torch already mapped;
    # the other order maps two runtimes and the second one sees no devices.
    if os.environ.get("SEMTOOLS_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    P = C.POINTER
    L.smt_last_error.restype = C.c_char_p
    L.smt_version.restype = C.c_char_p
    L.smt_device_count.restype = i32
    L.smt_ctx_create.argtypes = [i32, P(vp)]
    L.smt_ctx_create_on_stream.argtypes = [i32, vp, P(vp)]
    L.smt_ctx_aux_stream.argtypes = [vp, P(vp)]
    L.smt_ctx_destroy.argtypes = [vp]
    L.smt_ctx_destroy.restype = None
    L.smt_ctx_synchronize.argtypes = [vp]
    L.smt_prof_enable.argtypes = [vp, i32]
    L.smt_prof_reset.argtypes = [vp]
    L.smt_prof_read.argtypes = [vp, C.c_char_p, P(u64), P(f64)]
    L.smt_model_create.argtypes = [vp, vp, u64, u32, i32, P(vp)]
    L.smt_model_create_from_file.argtypes = [vp, C.c_char_p, u64, u64, u32, i32, P(vp)]
    L.smt_model_create_from_device.argtypes = [vp, vp, u64, u32, i32, P(vp)]
    L.smt_model_destroy.argtypes = [vp]
    L.smt_model_destroy.restype = None
    L.smt_embed.argtypes = [vp, vp, vp, u64, u32, vp, vp, P(u64)]
    L.smt_embed_device.argtypes = [vp, vp, vp, u64, u32, vp]
    L.smt_corpus_create.argtypes = [vp, u32, u64, P(vp)]
    L.smt_corpus_from_device.argtypes = [vp, vp, u64, u32, P(vp)]
    L.smt_corpus_destroy.argtypes = [vp]
    L.smt_corpus_destroy.restype = None
    L.smt_corpus_append_host.argtypes = [vp, vp, u64, P(u64)]
    L.smt_corpus_write_rows.argtypes = [vp, u64, vp, u64]
    L.smt_corpus_prepack.argtypes = [vp, C.c_int]
    L.smt_corpus_image_bytes.argtypes = [vp]
    L.smt_corpus_image_bytes.restype = u64
    L.smt_corpus_read_rows.argtypes = [vp, u64, u64, vp]
    L.smt_corpus_truncate.argtypes = [vp, u64]
    L.smt_corpus_rows.argtypes = [vp]
    L.smt_corpus_rows.restype = u64
    L.smt_corpus_dim.argtypes = [vp]
    L.smt_corpus_dim.restype = u32
    L.smt_corpus_save.argtypes = [vp, C.c_char_p]
    L.smt_corpus_load.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_corpus_append_to_file.argtypes = [vp, C.c_char_p, u64]
    L.smt_search.argtypes = [vp, vp, u32, u32, f64, i32, vp, u32, u64, vp, vp, vp, u64]
    L.smt_search_topk_device.argtypes = [vp, vp, u32, u32, u64, vp, vp]
    L.smt_search_topk_device_ex.argtypes = [vp, vp, u32, u32, u64, vp, vp, vp]
    L.smt_merge_topk.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp]
    L.smt_merge_topk_device.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp, vp]
    L.smt_merge_topk_packed_device.argtypes = [vp, vp, u32, u32, u32, u32, vp]
    L.smt_ivfpq_build.argtypes = [vp, P(SmtIvfPqParams), P(vp)]
    L.smt_ivfpq_destroy.argtypes = [vp]
    L.smt_ivfpq_destroy.restype = None
    L.smt_ivfpq_search.argtypes = [vp, vp, u32, u32, u32, u32, u64, vp, vp, vp, u64]
    L.smt_ivfpq_search_device.argtypes = [vp, vp, u32, u32, u32, u32, u64, vp, vp]
    L.smt_ivfpq_info.argtypes = [vp, P(u64), P(u32), P(u64), P(f64)]
    L.smt_ivfpq_list_sizes.argtypes = [vp, vp]
    L.smt_ivfpq_append.argtypes = [vp, P(u64)]
    L.smt_ivfpq_save.argtypes = [vp, C.c_char_p]
    L.smt_ivfpq_load.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_set_tuning.argtypes = [vp, C.c_char_p, C.c_int64]
    L.smt_fnv1a_hash.argtypes = [C.c_char_p, u64]
    L.smt_fnv1a_hash.restype = u64
    L.smt_line_embedding_id.argtypes = [C.c_char_p, C.c_int32]
    L.smt_line_embedding_id.restype = u64
    L.smt_doc_meta_id.argtypes = [C.c_char_p]
    L.smt_doc_meta_id.restype = u64
    L.smt_ctx_uncertain_count.argtypes = [vp, P(u64), i32]
    L.smt_debug_batched_scores.argtypes = [vp, vp, C.c_uint32, u64, C.c_uint32, vp]
    # ---- groups of GPUs
    L.smt_init.argtypes = [P(i32), i32]
    L.smt_shutdown.argtypes = []
    L.smt_default_group.argtypes = []
    L.smt_default_group.restype = vp
    L.smt_group_create.argtypes = [P(i32), i32, P(vp)]
    L.smt_group_create_logical.argtypes = [i32, i32, P(vp)]
    L.smt_group_unique_id.argtypes = [vp]
    L.smt_group_create_rank.argtypes = [i32, i32, i32, vp, P(vp)]
    L.smt_group_destroy.argtypes = [vp]
    L.smt_group_destroy.restype = None
    L.smt_group_info.argtypes = [vp, P(i32), P(i32), P(i32), P(i32), P(i32)]
    L.smt_group_ctx.argtypes = [vp, i32]
    L.smt_group_ctx.restype = vp
    L.smt_group_synchronize.argtypes = [vp]
    L.smt_group_barrier.argtypes = [vp]
    try:   # (a test hook: an older build of the library under tools/ab_* A/B runs does not have it)
        L.smt_debug_range_sets.argtypes = [vp, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
        L.smt_debug_group_fail_next.argtypes = [vp, i32, i32]
        L.smt_debug_deliveries.argtypes = [vp, P(C.c_uint64)]
    except AttributeError:
        pass
    L.smt_sharded_corpus_append_to_file_ex.argtypes = [vp, C.c_char_p, u64, u64, i32]
    L.smt_group_set_transport.argtypes = [vp, i32]
    L.smt_group_transport.argtypes = [vp]
    L.smt_sharded_corpus_from_host.argtypes = [vp, vp, u64, u32, P(vp)]
    L.smt_sharded_corpus_from_device.argtypes = [vp, P(vp), P(u64), u32, P(vp)]
    L.smt_sharded_corpus_load.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_sharded_corpus_save.argtypes = [vp, C.c_char_p]
    L.smt_sharded_corpus_destroy.argtypes = [vp]
    L.smt_sharded_corpus_destroy.restype = None
    L.smt_sharded_corpus_rows.argtypes = [vp]
    L.smt_sharded_corpus_rows.restype = u64
    L.smt_sharded_corpus_rank_rows.argtypes = [vp, vp]
    L.smt_sharded_corpus_shard.argtypes = [vp, i32, P(vp), P(u64), P(u64)]
    L.smt_sharded_corpus_append_host.argtypes = [vp, vp, u64, P(u64)]
    L.smt_sharded_search.argtypes = [vp, vp, u32, u32, f64, i32, vp, u32, vp, vp, vp, u64]
    L.smt_sharded_search_topk_device.argtypes = [vp, P(vp), u32, u32, P(vp)]
    L.smt_sharded_search_topk_device_ex.argtypes = [vp, P(vp), u32, u32, P(vp), P(vp)]
    L.smt_sharded_ivfpq_build.argtypes = [vp, P(SmtIvfPqParams), i32, P(vp)]
    L.smt_sharded_ivfpq_destroy.argtypes = [vp]
    L.smt_sharded_ivfpq_destroy.restype = None
    L.smt_sharded_ivfpq_shard.argtypes = [vp, i32]
    L.smt_sharded_ivfpq_shard.restype = vp
    L.smt_sharded_ivfpq_search.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp, u64]
    L.smt_group_from_ctx.argtypes = [vp, P(vp)]
    L.smt_sharded_corpus_create.argtypes = [vp, u32, P(vp)]
    L.smt_sharded_corpus_load_layout.argtypes = [vp, C.c_char_p, vp, vp, u64, P(vp)]
    L.smt_sharded_corpus_layout.argtypes = [vp, vp, vp, u64]
    L.smt_sharded_corpus_layout.restype = u64
    L.smt_sharded_corpus_append_to_file.argtypes = [vp, C.c_char_p, u64]
    L.smt_sharded_corpus_read_rows.argtypes = [vp, u64, u64, vp]
    L.smt_sharded_corpus_write_rows.argtypes = [vp, u64, vp, u64]
    L.smt_sharded_model_create.argtypes = [vp, vp, u64, u32, i32, P(vp)]
    L.smt_sharded_model_create_from_file.argtypes = [vp, C.c_char_p, u64, u64, u32, i32, P(vp)]
    L.smt_sharded_model_destroy.argtypes = [vp]
    L.smt_sharded_model_destroy.restype = None
    L.smt_sharded_embed.argtypes = [vp, vp, vp, u64, u32, vp, vp, P(u64)]
    L.smt_sharded_ivfpq_save.argtypes = [vp, C.c_char_p]
    L.smt_sharded_ivfpq_load.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_sharded_ivfpq_append.argtypes = [vp, P(u64)]
    L.smt_sharded_ivfpq_info.argtypes = [vp, P(u64), P(u32), P(u64)]
    # ---- host layer (include/semtools_host.h)
    cpp = P(C.c_char_p)
    L.smt_host_group_from_spec.argtypes = [C.c_char_p, P(vp)]
    L.smt_host_model_create_group.argtypes = [vp, vp, u64, i32, i32, C.c_char_p, C.c_char_p, TOKENIZE_CB, vp, u32, u32, P(vp)]
    L.smt_host_model_from_dir_group.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_host_workspace_use_group.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_workspace_status_group.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_workspace_prune_group.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_model_create.argtypes = [vp, vp, u64, i32, i32, C.c_char_p, C.c_char_p, TOKENIZE_CB, vp, u32, u32, P(vp)]
    L.smt_host_model_from_dir.argtypes = [vp, C.c_char_p, P(vp)]
    L.smt_host_model_destroy.argtypes = [vp]
    L.smt_host_model_destroy.restype = None
    L.smt_host_encode.argtypes = [vp, cpp, u64, u32, vp]
    L.smt_host_search_files.argtypes = [vp, C.c_char_p, cpp, u64, u64, u64, f64, i32, i32, i32, P(vp)]
    L.smt_host_search_content.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, u64, u64, f64, i32, i32, i32, P(vp)]
    L.smt_host_search_workspace.argtypes = [vp, C.c_char_p, cpp, u64, u64, u64, f64, i32, C.c_char_p, i32, i32, P(vp)]
    L.smt_host_session_open.argtypes = [vp, cpp, u64, i32, P(vp)]
    L.smt_host_session_search.argtypes = [vp, cpp, u64, u64, u64, f64, i32, i32, P(vp)]
    L.smt_host_session_lines.argtypes = [vp]
    L.smt_host_session_lines.restype = u64
    L.smt_host_session_close.argtypes = [vp]
    L.smt_host_session_close.restype = None
    L.smt_host_workspace_use.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_workspace_status.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_workspace_prune.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_workspace_reembed.argtypes = [vp, C.c_char_p, i32, P(vp)]
    L.smt_host_free.argtypes = [vp]
    L.smt_host_free.restype = None
    L.smt_host_tokenizer_load.argtypes = [C.c_char_p, P(vp)]
    L.smt_host_tokenizer_free.argtypes = [vp]
    L.smt_host_tokenizer_free.restype = None
    L.smt_host_tokenizer_encode.argtypes = [vp, C.c_char_p, vp, u64, P(u64)]
    L.smt_host_tokenizer_info.argtypes = [vp, P(u64), P(C.c_int64), P(u64)]
    L.smt_host_timing_json.argtypes = []
    L.smt_host_timing_json.restype = vp
    L.smt_host_format_float.argtypes = [f64, i32]
    L.smt_host_format_float.restype = vp
    L.smt_host_split_lines.argtypes = [C.c_char_p]
    L.smt_host_split_lines.restype = vp
    L.smt_host_to_lowercase.argtypes = [C.c_char_p]
    L.smt_host_to_lowercase.restype = vp
    _lib = L
    return L


def check(rc, allow=()):
    if rc != SMT_OK and rc not in allow:
        raise SmtError(rc, lib().smt_last_error().decode(errors="replace"))
    return rc


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
