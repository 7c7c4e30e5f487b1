"""ctypes loader for libprobly_search_amd.so (C ABI: include/probly_search_amd.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PS_SO") or os.path.join(_HERE, "csrc", "libprobly_search_amd.so")  # PS_SO: tuning builds


class LibraryNotBuilt(ImportError):
    pass


class PsError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, "%s (ps_status %d)" % (msg, status))
        self.status = status


PS_OK, PS_EINVAL, PS_ENOMEM, PS_EHIP, PS_EUNSUPPORTED, PS_ENODEVICE, PS_ERCCL = range(7)
PS_COMM_ID_BYTES = 128
PS_ABI_VERSION = 6  # include/probly_search_amd.h


class Str(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_size_t)]


class Result(C.Structure):
    _fields_ = [("key", C.c_uint64), ("score", C.c_double)]


class TermData(C.Structure):
    _fields_ = [("query_term_index", C.c_size_t), ("query_term", Str), ("query_term_expanded", Str),
                ("query_terms_len", C.c_size_t)]


class FieldDetailsC(C.Structure):
    _fields_ = [("sum", C.c_uint64), ("avg", C.c_double)]


class FieldData(C.Structure):
    _fields_ = [("fields_boost", C.POINTER(C.c_double)), ("n_boost", C.c_size_t),
                ("fields", C.POINTER(FieldDetailsC)), ("n_fields", C.c_size_t)]


class DocumentPointer(C.Structure):
    _fields_ = [("details_key", C.c_uint64), ("term_frequency", C.POINTER(C.c_uint32))]


class DocumentDetails(C.Structure):
    _fields_ = [("key", C.c_uint64), ("field_length", C.POINTER(C.c_uint32))]


BEFORE_EACH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TermData), C.c_size_t, C.c_size_t, C.c_void_p,
                             C.POINTER(C.c_void_p))
SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(DocumentPointer), C.POINTER(DocumentDetails),
                       C.c_uint64, C.POINTER(FieldData), C.POINTER(TermData), C.POINTER(C.c_double))
FINALIZE_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(Result), C.c_size_t)
DROP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class ScoreCallbacks(C.Structure):
    _fields_ = [("before_each", BEFORE_EACH_FN), ("score", SCORE_FN), ("finalize", FINALIZE_FN),
                ("drop_memory", DROP_FN), ("user", C.c_void_p)]


class ScorerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("bm25_k1", C.c_double), ("bm25_b", C.c_double),
                ("callbacks", C.POINTER(ScoreCallbacks))]


class UpdateStats(C.Structure):
    _fields_ = [("mode", C.c_int32), ("trie_refrozen", C.c_int32), ("docs_added", C.c_uint64),
                ("docs_removed", C.c_uint64), ("postings_uploaded", C.c_uint64), ("bytes_uploaded", C.c_uint64),
                ("delta_layers", C.c_uint64), ("delta_postings", C.c_uint64), ("host_ms", C.c_double),
                ("device_ms", C.c_double)]


class KernelTimes(C.Structure):
    _fields_ = [("score_ms", C.c_double), ("rows_ms", C.c_double), ("launches", C.c_uint64),
                ("score_kernel", C.c_char * 96), ("score_busy_ms", C.c_double)]


class WorkCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "launches", "items", "items_run", "postings_scanned", "postings_reached_lookups", "lookups_row", "lookups_cell",
        "lookups_probe", "lookup_hits", "offers", "k1_items", "k1_postings", "k1_row_slices", "results", "rows_built", "rows_used",
        "bytes_touched", "z_postings_scanned", "z_lookup_hits")]


class SnapshotInfo(C.Structure):
    _fields_ = [("fields_num", C.c_uint32), ("tile_docs", C.c_uint32), ("n_docs", C.c_uint64),
                ("n_terms", C.c_uint64), ("n_postings", C.c_uint64), ("n_pointers", C.c_uint64),
                ("n_table_entries", C.c_uint64), ("device_bytes", C.c_uint64), ("device", C.c_int32),
                ("max_layers", C.c_int32), ("n_ids", C.c_uint64), ("tiles_cap", C.c_uint32),
                ("delta_layers", C.c_uint32), ("delta_postings", C.c_uint64)]


class BatchStats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_plan_entries", C.c_uint64), ("postings_visited", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("plan_ms", C.c_double), ("h2d_ms", C.c_double),
                ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("score_kernel_ms", C.c_double),
                ("total_ms", C.c_double), ("layout_bytes", C.c_uint64), ("dense_rows", C.c_uint32),
                ("dense_rows_built", C.c_uint32), ("device_planned", C.c_uint32), ("bounds_recomputed", C.c_uint32)]


class PlanEntry(C.Structure):
    _fields_ = [("post_off", C.c_uint64), ("len", C.c_uint32), ("tbl_off", C.c_uint32), ("shift", C.c_uint32),
                ("qterm", C.c_uint32), ("idf", C.c_double), ("boost", C.c_double), ("node", C.c_uint32),
                ("qterm_index", C.c_uint32), ("bm_off", C.c_uint32), ("layer", C.c_uint32)]


class HostCsr(C.Structure):
    _fields_ = [("doc", C.POINTER(C.c_uint32)), ("tf", C.POINTER(C.c_uint32)), ("fl", C.POINTER(C.c_uint32)),
                ("table", C.POINTER(C.c_uint32)), ("keys", C.POINTER(C.c_uint64)), ("avg", C.POINTER(C.c_double)),
                ("plane_stride", C.c_uint64), ("alive", C.POINTER(C.c_uint32))]


TOKENIZER_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                           C.c_size_t, C.c_void_p)

# every symbol include/probly_search_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ps_last_error": (C.c_char_p, []),
    "ps_free": (None, [_P]),
    "ps_results_split": (None, [_P, C.c_size_t, _P, _P]),
    "ps_device_count": (C.c_int, []),
    "ps_abi_version": (C.c_uint32, []),
    "ps_set_option": (C.c_int, [C.c_char_p, C.c_uint32]),
    "ps_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32)]),
    "ps_index_new": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "ps_index_new_with_capacity": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "ps_index_free": (None, [_P]),
    "ps_index_add_document": (C.c_int, [_P, C.c_uint64, C.POINTER(Str), C.POINTER(C.c_size_t), _P, _P]),
    "ps_index_add_documents_flat": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "ps_index_add_documents_flat_gpu": (C.c_int, [_P, C.c_size_t, _P, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "ps_index_remove_document": (C.c_int, [_P, C.c_uint64]),
    "ps_index_vacuum": (C.c_int, [_P]),
    "ps_index_fields_len": (C.c_size_t, [_P]),
    "ps_index_docs_len": (C.c_size_t, [_P]),
    "ps_index_field_details": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "ps_index_doc_field_length": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ps_index_count_nodes": (C.c_size_t, [_P]),
    "ps_index_live_pointers": (C.c_size_t, [_P]),
    "ps_index_children": (C.c_long, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]),
    "ps_index_count_documents": (C.c_long, [_P, C.c_char_p, C.c_size_t]),
    "ps_index_expand_term": (C.c_size_t, [_P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]),
    "ps_index_snapshot": (C.c_int, [_P, C.c_int, C.c_uint32, C.POINTER(_P)]),
    "ps_snapshot_free": (None, [_P]),
    "ps_snapshot_save": (C.c_int, [_P, C.c_char_p]),
    "ps_snapshot_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "ps_snapshot_get_info": (C.c_int, [_P, C.POINTER(SnapshotInfo)]),
    "ps_snapshot_query": (C.c_int, [_P, C.POINTER(ScorerDesc), C.c_char_p, C.c_size_t, C.POINTER(C.c_double),
                                    C.c_size_t, _P, _P, C.c_size_t, C.POINTER(C.POINTER(Result)),
                                    C.POINTER(C.c_size_t)]),
    "ps_index_query": (C.c_int, [_P, C.POINTER(ScorerDesc), C.c_char_p, C.c_size_t, C.POINTER(C.c_double),
                                 C.c_size_t, _P, _P, C.c_size_t, C.POINTER(C.POINTER(Result)),
                                 C.POINTER(C.c_size_t)]),
    "ps_snapshot_query_batch": (C.c_int, [_P, C.POINTER(ScorerDesc), C.POINTER(Str), C.c_size_t,
                                          C.POINTER(C.c_double), C.c_size_t, _P, _P, C.c_size_t,
                                          C.POINTER(C.POINTER(Result)), C.POINTER(C.POINTER(C.c_size_t))]),
    "ps_snapshot_query_batch_device": (C.c_int, [_P, C.POINTER(ScorerDesc), C.POINTER(Str), C.c_size_t,
                                                 C.POINTER(C.c_double), C.c_size_t, _P, _P, C.c_size_t, _P, _P, _P,
                                                 _P]),
    "ps_snapshot_query_batch_device_flat": (C.c_int, [_P, C.POINTER(ScorerDesc), _P, _P, C.c_size_t,
                                                      C.POINTER(C.c_double), C.c_size_t, _P, _P, C.c_size_t, _P, _P,
                                                      _P, _P]),
    "ps_snapshot_query_batch_device_planned_flat": (C.c_int, [_P, C.POINTER(ScorerDesc), _P, _P, C.c_size_t,
                                                              C.POINTER(C.c_double), C.c_size_t, C.c_size_t, _P, _P,
                                                              _P, _P]),
    "ps_snapshot_plan_device": (C.c_int, [_P, C.POINTER(ScorerDesc), _P, _P, C.c_size_t,
                                          C.POINTER(C.POINTER(PlanEntry)), C.POINTER(C.c_size_t),
                                          C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]),
    "ps_snapshot_last_stats": (C.c_int, [_P, C.POINTER(BatchStats)]),
    "ps_snapshot_kernel_times": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "ps_snapshot_kernel_breakdown": (C.c_int, [_P, C.POINTER(KernelTimes), C.c_int]),
    "ps_snapshot_work_counters": (C.c_int, [_P, C.POINTER(WorkCounters), C.c_int]),
    "ps_index_snapshot_ex": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "ps_snapshot_update": (C.c_int, [_P, _P, C.POINTER(UpdateStats)]),
    "ps_index_snapshot_multi": (C.c_int, [_P, C.POINTER(C.c_int), C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "ps_snapshot_plan_ahead_flat": (C.c_int, [_P, C.POINTER(ScorerDesc), _P, _P, C.c_size_t, C.POINTER(C.c_int)]),
    "ps_comm_get_unique_id": (C.c_int, [_P]),
    "ps_comm_init_rank": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "ps_comm_free": (None, [_P]),
    "ps_comm_world_size": (C.c_int, [_P]),
    "ps_comm_rccl_path": (C.c_char_p, []),
    "ps_comm_rank": (C.c_int, [_P]),
    "ps_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "ps_topk_block_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "ps_snapshot_query_batch_allgather_flat": (C.c_int, [_P, _P, C.POINTER(ScorerDesc), _P, _P, C.c_size_t,
                                                         C.POINTER(C.c_double), C.c_size_t, _P, _P, C.c_size_t, _P,
                                                         _P, _P]),
    "ps_snapshot_plan": (C.c_int, [_P, C.POINTER(ScorerDesc), C.c_char_p, C.c_size_t, _P, _P,
                                   C.POINTER(C.POINTER(PlanEntry)), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ps_snapshot_host_csr": (C.c_int, [_P, C.POINTER(HostCsr)]),
    "ps_keytable_new": (C.c_int, [C.POINTER(_P)]),
    "ps_keytable_free": (None, [_P]),
    "ps_keytable_len": (C.c_size_t, [_P]),
    "ps_keytable_intern": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "ps_keytable_intern_flat": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "ps_keytable_find": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "ps_keytable_key": (C.c_int, [_P, C.c_uint64, _P]),
    "ps_keytable_resolve": (C.c_int, [_P, _P, C.c_size_t, _P]),
}

_lib = None


def lib_path():
    return _SO


def _preload_hip_runtime():
    """One HIP runtime per process.  The library needs `libamdhip64.so.7`; PyTorch-ROCm bundles
    its own copy (same SONAME) and loads it by file name, so if ours pulled /opt/rocm's first,
    a later `import torch` would bring up a second runtime and find no GPUs.  When torch is
    installed we therefore load ITS libamdhip64 first (without importing torch); the dynamic
    linker then binds our NEEDED entry to it by SONAME.  PS_HIP_RUNTIME=system opts out."""
    if os.environ.get("PS_HIP_RUNTIME", "auto") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def load():
    """Loads the HIP extension.  Raises LibraryNotBuilt (never falls back to anything)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise LibraryNotBuilt("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % _SO)
        _preload_hip_runtime()
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.ps_abi_version() != PS_ABI_VERSION:  # the ctypes structs above mirror ONE header version
            raise LibraryNotBuilt("%s was built for ABI version %d, this binding mirrors version %d: rebuild (make -C csrc)" % (
                _SO, L.ps_abi_version(), PS_ABI_VERSION))
        _lib = L
    return _lib


def check(status):
    if status != PS_OK:
        msg = load().ps_last_error()
        raise PsError(status, msg.decode("utf-8", "replace") if msg else "error")
