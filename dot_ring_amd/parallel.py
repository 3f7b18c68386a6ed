"""Multi-GPU sharding of the Ring-VRF hot path: one process per GPU, RCCL over xGMI, no PyTorch.

The path shards two ways.  Across independent proofs there is no collective at all: rank g proves and verifies
its own slice.  Across the bases of ONE large MSM (the reference's KZG.commit, dot_ring/ring_proof/pcs/kzg.py:152-175)
each rank reduces its shard of (base, scalar) pairs to one G1 point, the points are all-gathered (97 bytes per rank —
latency-bound on xGMI, nowhere near the per-link bandwidth) and every rank folds them with the group law.  Point addition
is not an RCCL reduction op, so this is an all-gather followed by a local fold, not an all-reduce.

`prove_batch_sharded` / `batch_verify_sharded` are the library form of the first way (BASELINE configs[4]): every rank calls them
with the SAME whole batch, rank g proves (verifies) proofs [g B / G, (g + 1) B / G) on its GPU, and the 784-byte proofs are
gathered — to every rank or to one — so that the caller holds all B of them, in order (the reference's process-sharded bench
hands its index ranges out and takes every result back the same way, tests/benchmark/bench_ring_proof.py:168-182).

Communicators (same small interface: rank, world, all_gather(bytes) -> list[bytes], barrier(), close()):

* `RcclComm`   — the product path: ncclAllGather through the C ABI (`dr_comm_*`, dot_ring_amd/csrc/capi_comm.hip).  The
                 ncclUniqueId travels from rank 0 to the others over a TCP socket on MASTER_ADDR (the launcher —
                 `python -m torch.distributed.run` or anything else that sets RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT —
                 is only a process starter; torch is never imported).
* `SocketComm` — the same exchange over plain TCP through rank 0: CPU tests, and rehearsals of N ranks on a box with fewer
                 GPUs (RCCL refuses two ranks on one device).
* `TorchComm`  — an existing torch.distributed group (gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes
import os
import socket
import struct
import time

from . import _native


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) slice of n items owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank / world size")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------------------------------------ rendezvous
def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world) as the launcher exported them (defaults: a single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _comm_endpoint() -> tuple[str, int]:
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("DOTRING_COMM_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29500")) + 1
    return addr, port


def _recv_exact(conn: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


class SocketComm:
    """All-gather through rank 0 over TCP (star).  Rank 0 listens on (addr, port); every other rank keeps one connection."""

    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None, timeout: float = 120.0):
        if world < 1 or not 0 <= rank < world:
            raise ValueError("bad rank / world size")
        d_addr, d_port = _comm_endpoint()
        self.rank, self.world = rank, world
        self._addr, self._port = addr or d_addr, port or d_port
        self._peers: list[socket.socket | None] = []
        self._up: socket.socket | None = None
        self._srv: socket.socket | None = None
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((self._addr, self._port))
            srv.listen(world)
            srv.settimeout(timeout)
            self._srv = srv
            peers: list[socket.socket | None] = [None] * world
            for _ in range(world - 1):
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                if not 0 < r < world or peers[r] is not None:
                    raise ConnectionError(f"unexpected rank {r} at the rendezvous")
                peers[r] = conn
            self._peers = peers
        else:
            deadline = time.monotonic() + timeout
            while True:
                try:
                    up = socket.create_connection((self._addr, self._port), timeout=timeout)
                    break
                except OSError:
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.05)
            up.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            up.sendall(struct.pack("<I", rank))
            self._up = up

    def all_gather(self, mine: bytes) -> list[bytes]:
        if self.world == 1:
            return [bytes(mine)]
        n = len(mine)
        if self.rank == 0:
            parts = [bytes(mine)] + [_recv_exact(self._peers[r], n) for r in range(1, self.world)]
            blob = b"".join(parts)
            for r in range(1, self.world):
                self._peers[r].sendall(blob)
            return parts
        self._up.sendall(mine)
        blob = _recv_exact(self._up, n * self.world)
        return [blob[i * n : (i + 1) * n] for i in range(self.world)]

    def broadcast(self, data: bytes | None, nbytes: int) -> bytes:
        """`data` of rank 0 to everyone (all ranks pass the length)."""
        if self.world == 1:
            return bytes(data)
        if self.rank == 0:
            for r in range(1, self.world):
                self._peers[r].sendall(data)
            return bytes(data)
        return _recv_exact(self._up, nbytes)

    def gather(self, mine: bytes, sizes: list) -> list | None:
        """Every rank's bytes (rank r sends sizes[r] of them; all ranks pass the same sizes) to rank 0: the list there, None
        elsewhere.  One hop per rank — half the bytes of all_gather on this star, and nothing travels back."""
        if self.world == 1:
            return [bytes(mine)]
        if len(mine) != sizes[self.rank]:
            raise ValueError("gather: this rank's payload does not have the announced size")
        if self.rank == 0:
            return [bytes(mine)] + [_recv_exact(self._peers[r], sizes[r]) for r in range(1, self.world)]
        if mine:
            self._up.sendall(mine)
        return None

    def barrier(self) -> None:
        self.all_gather(b"\0")

    def close(self) -> None:
        for s in [self._up, self._srv] + [p for p in self._peers if p is not None]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._up = self._srv = None
        self._peers = []


class RcclComm:
    """ncclAllGather over the ranks of a job, one rank = one GPU context (`dr_comm_*`).  Creation is a collective."""

    def __init__(self, ctx: "_native.Context", rank: int, world: int, bootstrap: SocketComm | None = None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.handle = None
        own_boot = bootstrap is None
        boot = bootstrap or SocketComm(rank, world)
        try:
            # rank 0 ALWAYS broadcasts: one status byte (0 = id follows, 1 = ncclGetUniqueId failed) + the 128-byte id, so a
            # failure on rank 0 reaches the peers as an exception instead of leaving them in recv
            uid = ctypes.create_string_buffer(_native.COMM_ID_BYTES)
            status, err = 0, ""
            if rank == 0:
                if _native.lib().dr_comm_unique_id(uid) != 0:
                    status, err = 1, _native.last_error()
            msg = boot.broadcast(bytes([status]) + uid.raw if rank == 0 else None, 1 + _native.COMM_ID_BYTES)
            if msg[0]:
                raise _native.DotRingHipError("rank 0 could not create the RCCL unique id" + (f": {err}" if err else ""))
            uid_bytes = msg[1:]
            handle = ctypes.c_void_p()
            rc = _native.lib().dr_comm_create(ctx.handle, uid_bytes, rank, world, ctypes.byref(handle))
            err = _native.last_error() if rc else ""
            if rc == 0:
                self.handle = handle          # owned from here on: close() / __del__ destroy it whatever happens below
            # every rank learns whether ncclCommInitRank succeeded everywhere before anyone enters a collective on it
            oks = boot.all_gather(bytes([1 if rc == 0 else 0]))
        finally:
            if own_boot:
                boot.close()
        if rc != 0:
            raise _native.DotRingHipError(f"ncclCommInitRank failed on rank {rank}: {err}")
        bad = [r for r, b in enumerate(oks) if not b[0]]
        if bad:
            self.close()
            raise _native.DotRingHipError(f"ncclCommInitRank failed on rank(s) {bad}")

    def rccl_ranks(self) -> int:
        """ranks RCCL itself counts in the communicator (ncclCommCount)"""
        n = ctypes.c_int(0)
        _native._check(_native.lib().dr_comm_count(self.handle, ctypes.byref(n)))
        return n.value

    def all_gather(self, mine: bytes) -> list[bytes]:
        n = len(mine)
        out = ctypes.create_string_buffer(max(1, n * self.world))
        _native._check(_native.lib().dr_comm_all_gather(self.handle, bytes(mine), n, out))
        return [out.raw[i * n : (i + 1) * n] for i in range(self.world)]

    def barrier(self) -> None:
        self.ctx.sync()
        self.all_gather(b"\0")

    def g1_msm_sharded_dev(self, srs, d_scalars, n_local: int, offset: int = 0) -> bytes | None:
        """The native fused path: local MSM -> ncclAllGather -> fold, one call (`dr_g1_msm_sharded_dev`)."""
        out, inf = ctypes.create_string_buffer(96), ctypes.c_int(0)
        _native._check(_native.lib().dr_g1_msm_sharded_dev(self.ctx.handle, self.handle, srs.handle, offset, d_scalars.ptr, n_local, out,
                                                           ctypes.byref(inf)))
        return None if inf.value else out.raw

    def close(self) -> None:
        if getattr(self, "handle", None):
            _native.lib().dr_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchComm:
    """An initialised torch.distributed process group (gloo on CPU, nccl = RCCL on GPUs) behind the same interface."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def all_gather(self, mine: bytes) -> list[bytes]:
        import torch

        if self.world == 1:
            return [bytes(mine)]
        on_gpu = self._dist.get_backend(self._group) == "nccl"
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        if on_gpu:
            t = t.cuda()
        gathered = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(gathered, t, group=self._group)
        return [bytes(g.cpu().numpy().tobytes()) for g in gathered]

    def barrier(self) -> None:
        if self.world > 1:
            self._dist.barrier(group=self._group)

    def close(self) -> None:
        pass


def make_comm(ctx=None, backend: str | None = None):
    """The communicator of this process as the launcher's environment describes it.  backend: "rccl" (default with a GPU
    context), "socket" (DOTRING_COMM=socket: ranks may share a GPU; also the CPU tests)."""
    rank, _, world = env_rank_world()
    backend = backend or os.environ.get("DOTRING_COMM", "rccl" if ctx is not None else "socket")
    if backend == "socket":
        return SocketComm(rank, world)
    if backend == "rccl":
        if ctx is None:
            raise ValueError("the RCCL communicator needs a GPU context")
        return RcclComm(ctx, rank, world)
    raise ValueError(f"unknown communicator backend {backend!r}")


# ------------------------------------------------------------------------------------------------ base-sharded MSM
def _pack_point(point: bytes | None) -> bytes:
    return (bytes(96) + b"\x01") if point is None else (bytes(point) + b"\x00")


def all_gather_points(point: bytes | None, comm=None) -> list:
    """All-gather one affine G1 point (96-byte BE record, None = infinity) from every rank."""
    if comm is None:
        comm = TorchComm()
    return [None if rec[96] else rec[:96] for rec in comm.all_gather(_pack_point(point))]


def combine_partials(point: bytes | None, comm=None) -> bytes | None:
    """Sum of every rank's partial MSM result; identical on all ranks.  `comm`: any communicator above (a
    torch.distributed group object is accepted for backward compatibility)."""
    if comm is not None and not hasattr(comm, "all_gather"):
        comm = TorchComm(comm)
    return _native.g1_sum(all_gather_points(point, comm))


def g1_msm_sharded(ctx, comm, srs_shard, d_scalars, n_local: int, offset: int = 0) -> bytes | None:
    """One MSM whose (base, scalar) pairs are sharded over the ranks of `comm`: `srs_shard` / `d_scalars` hold this rank's
    n_local pairs in HBM.  Runs the local MSM on the GPU, exchanges one point per rank, folds; the same 96 bytes (or None
    for infinity) on every rank.  With an RcclComm the whole thing is one native call."""
    if isinstance(comm, RcclComm):
        return comm.g1_msm_sharded_dev(srs_shard, d_scalars, n_local, offset)
    part = ctx.g1_msm_dev(srs_shard, d_scalars, n_local, offset) if n_local else None
    return combine_partials(part, comm)


# ------------------------------------------------------------------------------------------------ proof-sharded batches
class DotRingShardError(_native.DotRingHipError):
    """a shard of a sharded call failed on some rank (the message names the rank and carries its error)"""


_HDR = struct.Struct("<BQ")          # status (0 ok, 1 failed), payload bytes
# what the last exchange of this process moved and how long its payload phase took (benchmarks read it; the header all_gather before
# it absorbs the wait for the slowest rank, so `payload_s` is transfer time, not straggling)
last_exchange = {"what": None, "payload_bytes": 0, "payload_s": 0.0, "header_s": 0.0}


def _exchange(comm, what: str, mine: bytes, failed: str | None, dst: int | None):
    """The variable-size gather every sharded call ends with.  First a 9-byte header per rank (status + size) travels by
    all_gather, so that a rank whose shard failed is seen by all of them BEFORE anyone waits for its payload — every rank then
    raises the first failing rank's error (ValueError for the reference's argument errors, DotRingShardError otherwise).  Then
    the payloads: to every rank (dst None) or to rank `dst` only.  Returns the payload per rank, or None off `dst`."""
    t0 = time.perf_counter()
    hdrs = [_HDR.unpack(h) for h in comm.all_gather(_HDR.pack(1 if failed else 0, 0 if failed else len(mine)))]
    t1 = time.perf_counter()
    bad = [r for r, (st, _) in enumerate(hdrs) if st]
    if bad:
        texts = comm.all_gather((failed or "").encode()[:400].ljust(400))
        text = texts[bad[0]].decode(errors="replace").strip()
        kind = ValueError if text.startswith("ValueError:") else DotRingShardError
        raise kind(f"sharded {what} failed on rank {bad[0]}: {text}")
    sizes = [n for _, n in hdrs]
    if dst == 0 and hasattr(comm, "gather"):
        parts = comm.gather(mine, sizes)
    else:
        width = max(sizes)
        parts = comm.all_gather(bytes(mine).ljust(width, b"\0")) if width else [b""] * comm.world
        parts = None if dst is not None and comm.rank != dst else [part[:n] for part, n in zip(parts, sizes)]
    last_exchange.update(what=what, payload_bytes=sum(sizes), payload_s=time.perf_counter() - t1, header_s=t1 - t0)
    return parts


def prove_batch_sharded(comm, vrf, alphas, additional_data, secret_keys, producer_keys, ring, ring_root=None, salts=None,
                        dst: int | None = None):
    """RingVRF.prove_batch of ONE batch over the ranks of `comm` (one process per GPU): every rank passes the same arguments,
    rank g proves proofs [g B / G, (g + 1) B / G) on its own GPU, the encoded proofs (784 bytes each) are gathered, and the
    result is the list of all B proofs in order — element i equals vrf.prove(alphas[i], ...) — on every rank (`dst` None), or
    on rank `dst` only (None elsewhere: half the traffic).  Ragged splits and empty shards (B < G) are fine.  No collective
    runs while the GPUs work; the one exchange is this gather (B x 784 bytes, 6.4 MB at 8192 proofs).

    `vrf` is the bound scheme, e.g. dot_ring_amd.RingVRF[Bandersnatch].  If any rank's shard raises, EVERY rank raises
    instead of waiting for proofs that never come.  Mirrors the process-sharded driver of the reference's bench
    (tests/benchmark/bench_ring_proof.py:168-182: index ranges out, all results back) over dot_ring/vrf/ring/vrf.py:185-209."""
    count = len(alphas)
    if not (len(additional_data) == len(secret_keys) == len(producer_keys) == count) or (salts is not None and len(salts) != count):
        raise ValueError("batch arguments must have equal lengths")
    lo, hi = shard_range(count, comm.rank, comm.world)
    blob, failed = b"", None
    if hi > lo:
        try:
            mine = vrf.prove_batch(alphas[lo:hi], additional_data[lo:hi], secret_keys[lo:hi], producer_keys[lo:hi], ring, ring_root,
                                   None if salts is None else salts[lo:hi])
            blob = vrf.encode_batch(mine)
        except Exception as exc:                                   # noqa: BLE001 — reported to every rank, raised by _exchange
            failed = f"{type(exc).__name__}: {exc}"
    parts = _exchange(comm, "prove_batch", blob, failed, dst)
    if parts is None:
        return None
    for r, part in enumerate(parts):
        a, b = shard_range(count, r, comm.world)
        if len(part) != 784 * (b - a):
            raise DotRingShardError(f"rank {r} sent {len(part)} bytes for {b - a} proofs")
    return vrf._from_encoded(b"".join(parts), count)


def batch_verify_sharded(comm, vrf, proofs, inputs, additional_data, ring, ring_root) -> bool:
    """RingVRF.batch_verify of ONE batch over the ranks of `comm`: every rank passes the same inputs; rank g verifies proofs
    [g B / G, (g + 1) B / G) — its own random linear combination, its own pairing equation — and the verdicts are AND-ed
    (one byte per rank).  `proofs` may be None on every rank but 0 (prove_batch_sharded(dst=0) left them there): rank 0 then
    broadcasts the encoded batch first.  True on every rank iff every proof verifies (dot_ring/vrf/ring/vrf.py:239-283)."""
    count = len(inputs)
    if len(additional_data) != count or (proofs is not None and len(proofs) != count):
        raise ValueError("batch arguments must have equal lengths")
    have = comm.all_gather(bytes([0 if proofs is None else 1]))
    lo, hi = shard_range(count, comm.rank, comm.world)
    if all(h[0] for h in have):
        mine = proofs[lo:hi]
    else:
        if not have[0][0]:
            raise ValueError("batch_verify_sharded: rank 0 must hold the proofs when another rank does not")
        whole = vrf.encode_batch(proofs) if comm.rank == 0 else None
        if hasattr(comm, "broadcast"):
            whole = comm.broadcast(whole, 784 * count)
        else:
            whole = comm.all_gather(whole if comm.rank == 0 else bytes(784 * count))[0]
        mine = vrf._from_encoded(whole[784 * lo : 784 * hi], hi - lo)
    ok, failed = True, None
    if hi > lo:
        try:
            ok = bool(vrf.batch_verify(mine, inputs[lo:hi], additional_data[lo:hi], ring, ring_root))
        except Exception as exc:                                   # noqa: BLE001
            ok, failed = False, f"{type(exc).__name__}: {exc}"
    return all(p[0] for p in _exchange(comm, "batch_verify", bytes([1 if ok else 0]), failed, None))
