"""semtools_amd -- MI355X-native core for the `semtools search` hot path.

Thin Python surface over libsemtools_hip.so (C ABI: include/semtools_hip.h).
Names mirror the reference's Rust API for this path (src/search/mod.rs,
src/workspace/store.rs) so tests read like the reference's own.
"""
from .core import Context, Corpus, Group, IvfPq, Model, PackedRanges, ShardedCorpus, ShardedIvfPq, ShardedModel, merge_topk  # noqa: F401
from ._lib import SmtError, SmtRange, MODE_DOCUMENTS, MODE_WORKSPACE, DIM  # noqa: F401
