"""Per-thread GPU contexts for the API layer (one process per GPU: the device is LOCAL_RANK).

A dr_ctx (one HIP stream + its scratch buffers) must not be used from two threads at once, so every Python thread
that touches the API gets its own context on the process's device.  The batched prover uses this to pipeline two
half-batches: while one thread hashes transcripts on the host (GIL held), the other thread's kernels run.
"""
from __future__ import annotations

import os
import threading

from . import _native

_local = threading.local()


def device_index() -> int:
    return int(os.environ.get("DOTRING_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def context() -> _native.Context:
    """The lazily created context of the calling thread.  Raises if the library or the GPU is missing — no CPU path."""
    ctx = getattr(_local, "ctx", None)
    if ctx is None:
        ctx = _native.Context(device_index())
        _local.ctx = ctx
    return ctx


def set_context(ctx: _native.Context | None) -> None:
    _local.ctx = ctx
