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


def device_ids() -> list:
    """The device set of this process: DOTRING_DEVICES = "0,1,2,3" makes RingVRF.prove_batch / batch_verify shard every batch over
    those GPUs in-process (dr_ringvrf_prove_batch_multi; an id may repeat: several contexts on one GPU).  Unset: the one device of
    device_index() — one process per GPU, the launcher's LOCAL_RANK."""
    spec = os.environ.get("DOTRING_DEVICES", "").strip()
    if not spec:
        return [device_index()]
    return [int(tok) for tok in spec.split(",") if tok.strip() != ""]


if len(device_ids()) > 1 and "DOTRING_HOST_THREADS" not in os.environ:
    # the library's worker pool (hashing between the GPU phases) is sized once per process: about 16 threads per device
    os.environ["DOTRING_HOST_THREADS"] = str(max(2, min(len(os.sched_getaffinity(0)), 16 * len(device_ids()))))


def context() -> _native.Context:
    """The lazily created context of the calling thread.  Raises if the library or the GPU is missing — no CPU path."""
    ctx = getattr(_local, "ctx", None)
    if ctx is None:
        ctx = _native.Context(device_ids()[0])
        _local.ctx = ctx
        with _created_lock:
            _created[:] = [r for r in _created if r() is not None]
            _created.append(weakref.ref(ctx))
    return ctx


def device_contexts() -> list:
    """One context per entry of device_ids() for the calling thread; entry 0 is context()."""
    ids = device_ids()
    have = getattr(_local, "device_ctxs", None)
    if have is None or len(have) != len(ids) or have[0] is not context() or any(not c.handle for c in have):
        have = [context()]
        for dev in ids[1:]:
            ctx = _native.Context(dev)
            with _created_lock:
                _created.append(weakref.ref(ctx))
            have.append(ctx)
        _local.device_ctxs = have
    return have


def contexts() -> list:
    """All live contexts created through context() (the calling thread's and those of the prover's helper threads): profiling
    tools sum their per-kernel timers."""
    with _created_lock:
        return [c for c in (r() for r in _created) if c is not None and c.handle]


def set_context(ctx: _native.Context | None) -> None:
    _local.ctx = ctx
