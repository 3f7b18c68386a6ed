"""The Python orchestration of the provers (custom PCS, DOTRING_NATIVE_HOST=0) is written as generators that `yield`
zero-argument callables doing ONLY GPU work (ctypes calls into libdotring_hip) and receive the callable's result back; `drive` runs
such a generator on the calling thread.  (Rounds 1 - 3 also ran several slices of a batch round-robin against a GPU worker thread:
measured +0..4 %, dropped with the native batch orchestration — DESIGN 8.)
"""
from __future__ import annotations


def drive(gen):
    """Run a prover generator to completion in the calling thread."""
    try:
        task = next(gen)
        while True:
            task = gen.send(task())
    except StopIteration as stop:
        return stop.value
