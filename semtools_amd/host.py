"""Python handles for the host layer (include/semtools_host.h): the reference's search module and
workspace commands, driven through the C ABI.  Names follow src/search/mod.rs / src/cmds/*."""
import ctypes as C

import numpy as np

from . import _lib as L

NAN = float("nan")


def _take_text(ptr):
    if not ptr:
        return ""
    s = C.string_at(ptr).decode("utf-8", errors="surrogateescape")
    L.lib().smt_host_free(ptr)
    return s


def _cstrs(items):
    arr = (C.c_char_p * max(len(items), 1))()
    for i, s in enumerate(items):
        arr[i] = s.encode()
    return arr


class StaticModel:
    """model2vec_rs::model::StaticModel: tokenizer + resident embedding table.

    tokenizer: "hash" (whitespace words hashed into the table), ("vocab", path, unk_token) or a
    Python callable text -> list of ids (e.g. tokenizers.Tokenizer(...).encode(t, add_special_tokens=False).ids)."""

    def __init__(self, ctx, table=None, tokenizer="hash", normalize=True, unk_id=None, median_len=5, model_dir=None):
        """ctx: a core.Context (one GPU) or a core.Group (the host layer then runs sharded over its GPUs)."""
        from .core import Group
        self.ctx = ctx
        on_group = isinstance(ctx, Group)
        self._h = C.c_void_p()
        self._cb = L.TOKENIZE_CB()
        lib = L.lib()
        if model_dir is not None:
            fn = lib.smt_host_model_from_dir_group if on_group else lib.smt_host_model_from_dir
            L.check(fn(ctx._h, str(model_dir).encode(), C.byref(self._h)))
            return
        table = np.ascontiguousarray(table, dtype=np.float32)
        kind, vocab, unk = 0, None, None
        if isinstance(tokenizer, tuple) and tokenizer[0] == "vocab":
            kind, vocab, unk = 1, str(tokenizer[1]).encode(), (tokenizer[2] or "").encode()
        elif callable(tokenizer):
            kind = 2

            def _cb(user, text, n, ids, cap, n_out):
                try:
                    out = tokenizer(C.string_at(text, n).decode("utf-8", errors="replace"))
                    n_out[0] = len(out)
                    for i in range(min(len(out), cap)):
                        ids[i] = out[i]
                    return 0
                except Exception:  # surfaced as SMT_E_INVALID by the library
                    return 1

            self._cb = L.TOKENIZE_CB(_cb)
        fn = lib.smt_host_model_create_group if on_group else lib.smt_host_model_create
        L.check(fn(ctx._h, L.np_ptr(table), table.shape[0], int(normalize), kind, vocab, unk,
                   self._cb, None, 0xFFFFFFFF if unk_id is None else int(unk_id), int(median_len), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_host_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_with_args(self, sentences, max_length=2048):
        out = np.empty((len(sentences), L.DIM), dtype=np.float32)
        L.check(L.lib().smt_host_encode(self._h, _cstrs(sentences), len(sentences), int(max_length or 0), L.np_ptr(out)))
        return out

    def encode_single(self, sentence):
        return self.encode_with_args([sentence], 512)[0]


def search_files(model, query, files, n_lines=3, top_k=3, max_distance=None, ignore_case=False, json=False, is_tty=False):
    """search_cmd's plain-files branch: returns exactly what `semtools search` prints to stdout."""
    out = C.c_void_p()
    L.check(L.lib().smt_host_search_files(model._h, query.encode(), _cstrs(files), len(files), n_lines, top_k,
                                          NAN if max_distance is None else max_distance, int(ignore_case), int(json),
                                          int(is_tty), C.byref(out)))
    return _take_text(out)


def search_content(model, query, content, filename="<stdin>", n_lines=3, top_k=3, max_distance=None, ignore_case=False,
                   json=False, is_tty=False):
    out = C.c_void_p()
    L.check(L.lib().smt_host_search_content(model._h, query.encode(), filename.encode(), content.encode(), n_lines, top_k,
                                            NAN if max_distance is None else max_distance, int(ignore_case), int(json),
                                            int(is_tty), C.byref(out)))
    return _take_text(out)


def search_with_workspace(model, query, files, workspace_name=None, n_lines=3, top_k=3, max_distance=None,
                          ignore_case=False, json=False, is_tty=False):
    out = C.c_void_p()
    L.check(L.lib().smt_host_search_workspace(model._h, query.encode(), _cstrs(files), len(files), n_lines, top_k,
                                              NAN if max_distance is None else max_distance, int(ignore_case),
                                              workspace_name.encode() if workspace_name else None, int(json), int(is_tty),
                                              C.byref(out)))
    return _take_text(out)


class Session:
    """Resident session: files embedded once, batches of queries answered against the resident corpus."""

    def __init__(self, model, files, ignore_case=False):
        self.model = model
        self._h = C.c_void_p()
        L.check(L.lib().smt_host_session_open(model._h, _cstrs(files), len(files), int(ignore_case), C.byref(self._h)))

    @property
    def lines(self):
        return int(L.lib().smt_host_session_lines(self._h))

    def search(self, queries, n_lines=3, top_k=3, max_distance=None, json=False, is_tty=False):
        outs = (C.c_void_p * max(len(queries), 1))()
        L.check(L.lib().smt_host_session_search(self._h, _cstrs(queries), len(queries), n_lines, top_k,
                                                NAN if max_distance is None else max_distance, int(json), int(is_tty), outs))
        return [_take_text(outs[i]) for i in range(len(queries))]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_host_session_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ws_fn(ctx, name):
    """ctx may be a core.Context or a core.Group: the *_group twin of the entry point for the latter."""
    from .core import Group
    return getattr(L.lib(), name + "_group" if isinstance(ctx, Group) else name)


def workspace_use(ctx, name, json=False):
    out = C.c_void_p()
    L.check(_ws_fn(ctx, "smt_host_workspace_use")(ctx._h if ctx is not None else None, name.encode(), int(json), C.byref(out)))
    return _take_text(out)


def workspace_status(ctx, name=None, json=False):
    out = C.c_void_p()
    L.check(_ws_fn(ctx, "smt_host_workspace_status")(ctx._h, name.encode() if name else None, int(json), C.byref(out)))
    return _take_text(out)


def workspace_prune(ctx, name=None, json=False):
    out = C.c_void_p()
    L.check(_ws_fn(ctx, "smt_host_workspace_prune")(ctx._h, name.encode() if name else None, int(json), C.byref(out)))
    return _take_text(out)


def workspace_reembed(model, name=None, json=False):
    """Re-create every stored vector of the workspace from its cached token ids with `model` (smt_host_workspace_reembed)."""
    out = C.c_void_p()
    L.check(L.lib().smt_host_workspace_reembed(model._h, name.encode() if name else None, int(json), C.byref(out)))
    return _take_text(out)


def format_float(value, mode):
    return _take_text(L.lib().smt_host_format_float(float(value), int(mode)))


def split_lines(content):
    parts = _take_text(L.lib().smt_host_split_lines(content.encode())).split("\x1f")
    return parts[1:1 + int(parts[0])]


def to_lowercase(text):
    return _take_text(L.lib().smt_host_to_lowercase(text.encode()))
