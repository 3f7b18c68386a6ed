"""Per-thread GPU contexts for the API layer (one process per GPU: the device is LOCAL_RANK).

A dr_ctx (one HIP stream + its scratch buffers) must not be used from two threads at once, so every Python thread
that touches the API gets its own context on the process's device.  The batched prover uses this to pipeline two
half-batches: while one thread hashes transcripts on the host (GIL held), the other thread's kernels run.
"""
from __future__ import annotations

import os
import threading
import weakref

from . import _native

_local = threading.local()
_created: list = []                 # weak references to every context handed out by context()
_created_lock = threading.Lock()


def device_index() -> int:
    return int(os.environ.get("DOTRING_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def context() -> _native.Context:
    """The lazily created context of the calling thread.  Raises if the library or the GPU is missing — no CPU path."""
    ctx = getattr(_local, "ctx", None)
    if ctx is None:
        ctx = _native.Context(device_index())
        _local.ctx = ctx
        with _created_lock:
            _created[:] = [r for r in _created if r() is not None]
            _created.append(weakref.ref(ctx))
    return ctx


def contexts() -> list:
    """All live contexts created through context() (the calling thread's and those of the prover's helper threads): profiling
    tools sum their per-kernel timers."""
    with _created_lock:
        return [c for c in (r() for r in _created) if c is not None and c.handle]


def set_context(ctx: _native.Context | None) -> None:
    _local.ctx = ctx
