"""Software pipelining of host hashing against GPU work, without GIL contention.

Prover code is written as generators that `yield` zero-argument callables doing ONLY GPU work (ctypes calls into
libdotring_hip, which release the GIL) and receive the callable's result back.  `drive` runs such a generator
synchronously.  `run_pipelined` runs several of them (slices of one batch) round-robin: every GPU callable is
executed by one dedicated worker thread that shares the caller's context, while the calling thread does the host
work (transcript hashing, scalar arithmetic) of the other slices.  The worker never runs Python-level host work and
the caller never touches the GPU while tasks are in flight, so a dr_ctx is still used by one thread at a time.
"""
from __future__ import annotations

import sys
import threading
from concurrent.futures import ThreadPoolExecutor

from . import runtime

_worker: ThreadPoolExecutor | None = None
_lock = threading.Lock()


def drive(gen):
    """Run a prover generator to completion in the calling thread."""
    try:
        task = next(gen)
        while True:
            task = gen.send(task())
    except StopIteration as stop:
        return stop.value


def _gpu_worker() -> ThreadPoolExecutor:
    global _worker
    with _lock:
        if _worker is None:
            _worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dotring-gpu")
        return _worker


def run_pipelined(gens):
    """Run prover generators concurrently (host work of one overlaps GPU work of the others); returns their values."""
    gens = list(gens)
    if len(gens) == 1:
        return [drive(gens[0])]
    ctx = runtime.context()
    worker = _gpu_worker()

    def bound(task):
        def call():
            runtime.set_context(ctx)          # the worker adopts the caller's context (thread-local slot)
            return task()
        return call

    # the worker must re-take the GIL between two GPU tasks while this thread is hashing: with CPython's default 5 ms
    # switch interval the GPU would idle that long after every task
    old_interval = sys.getswitchinterval()
    sys.setswitchinterval(min(old_interval, 2e-4))
    try:
        return _round_robin(gens, worker, bound)
    finally:
        sys.setswitchinterval(old_interval)


def _round_robin(gens, worker, bound):
    results = [None] * len(gens)
    pending = {}
    for i, g in enumerate(gens):
        try:
            pending[i] = worker.submit(bound(next(g)))
        except StopIteration as stop:
            results[i] = stop.value
    while pending:
        for i in list(pending):
            value = pending[i].result()
            try:
                pending[i] = worker.submit(bound(gens[i].send(value)))
            except StopIteration as stop:
                results[i] = stop.value
                del pending[i]
    return results
