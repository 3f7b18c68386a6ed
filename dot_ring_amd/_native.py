"""ctypes binding of libdotring_hip.so (include/dotring_hip.h) — the only door from Python to the GPU kernels.

No CPU fallback exists: if the shared library is missing, or there is no gfx950 device, the calls raise.
Error mapping follows the reference's exception types at these seams (ValueError / MemoryError).
"""
from __future__ import annotations

import array
import ctypes
import itertools
import threading
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdotring_hip.so")

DR_OK, DR_ERR_INVALID, DR_ERR_NOMEM, DR_ERR_DEVICE, DR_ERR_NOTSQUARE = 0, -1, -2, -3, -4


class DotRingHipError(RuntimeError):
    """HIP runtime failure, or no usable MI355X device."""


_lib = None

# name -> (restype, argtypes); mirrors include/dotring_hip.h one to one
_PROTOTYPES = {
    "dr_version": (c_char_p, []),
    "dr_last_error": (c_char_p, []),
    "dr_device_count": (c_int, []),
    "dr_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "dr_ctx_destroy": (None, [c_void_p]),
    "dr_ctx_sync": (c_int, [c_void_p]),
    "dr_dev_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "dr_dev_free": (c_int, [c_void_p, c_void_p]),
    "dr_dev_upload": (c_int, [c_void_p, c_void_p, c_char_p, c_size_t]),
    "dr_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "dr_prof_enable": (c_int, [c_void_p, c_int]),
    "dr_prof_reset": (c_int, [c_void_p]),
    "dr_prof_get": (c_int, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int)]),
    "dr_bsn_scalar_mul_batch": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_void_p]),
    "dr_bsn_scalar_mul_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_bsn_msm": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_void_p]),
    "dr_bsn_msm_groups": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_size_t, c_void_p]),
    "dr_bsn_encode_to_curve_batch": (c_int, [c_void_p, c_char_p, c_size_t, c_void_p]),
    "dr_fr_sqrt": (c_int, [c_char_p, c_void_p]),
    "dr_fr_ops_selftest": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_void_p, c_void_p]),
    "dr_srs_load": (c_int, [c_void_p, c_char_p, c_size_t, POINTER(c_void_p)]),
    "dr_srs_synthetic": (c_int, [c_void_p, c_char_p, c_uint, c_size_t, POINTER(c_void_p)]),
    "dr_srs_powers": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, POINTER(c_void_p)]),
    "dr_g2_mul": (c_int, [c_char_p, c_char_p, c_char_p]),
    "dr_srs_precompute": (c_int, [c_void_p, c_void_p, c_int]),
    "dr_srs_table_info": (c_int, [c_void_p, c_size_t, c_size_t, POINTER(c_int)]),
    "dr_srs_download": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "dr_srs_destroy": (None, [c_void_p]),
    "dr_srs_size": (c_size_t, [c_void_p]),
    "dr_g1_msm": (c_int, [c_void_p, c_void_p, c_size_t, c_char_p, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_g1_msm_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_g1_msm_batch": (c_int, [c_void_p, c_void_p, c_char_p, c_size_t, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_g1_msm_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_g1_msm_points": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_g1_sum": (c_int, [c_char_p, c_size_t, c_void_p, POINTER(c_int)]),
    "dr_pairing_check": (c_int, [c_char_p, c_char_p, c_size_t, POINTER(c_int)]),
    "dr_g1_compress": (c_int, [c_char_p, c_int, c_void_p]),
    "dr_g1_decompress": (c_int, [c_char_p, c_void_p, POINTER(c_int)]),
    "dr_g1_serialize_check": (c_int, [c_char_p]),
    "dr_ring_prover_create": (c_int, [c_void_p, c_void_p, c_uint, c_uint, c_char_p, c_char_p, c_char_p, c_char_p, POINTER(c_void_p)]),
    "dr_ring_prover_destroy": (None, [c_void_p]),
    "dr_ring_prover_root": (c_int, [c_void_p, c_void_p, POINTER(c_int)]),
    "dr_ring_prover_fixed_coeffs": (c_int, [c_void_p, c_void_p]),
    "dr_ring_prove_witness": (c_int, [c_void_p, c_size_t, c_void_p, c_char_p, c_char_p, c_void_p, c_void_p, POINTER(c_int)]),
    "dr_ring_prove_quotient": (c_int, [c_void_p, c_size_t, c_char_p, c_void_p, POINTER(c_int)]),
    "dr_ring_prove_evals": (c_int, [c_void_p, c_size_t, c_char_p, c_void_p]),
    "dr_ring_prove_openings": (c_int, [c_void_p, c_size_t, c_char_p, c_void_p, POINTER(c_int)]),
    "dr_ctx_scratch_residue": (c_int, [c_void_p, POINTER(ctypes.c_uint64)]),
    "dr_ring_prover_wipe": (c_int, [c_void_p]),
    "dr_ring_prover_residue": (c_int, [c_void_p, POINTER(ctypes.c_uint64)]),
    "dr_ntt": (c_int, [c_void_p, c_void_p, c_uint, c_size_t, c_char_p, c_char_p]),
    "dr_ntt_dev": (c_int, [c_void_p, c_void_p, c_uint, c_size_t, c_char_p, c_char_p]),
}


class VrfSuiteStruct(ctypes.Structure):
    """dr_vrf_suite (include/dotring_hip.h)."""
    _fields_ = [("suite_id", c_char_p), ("suite_id_len", c_size_t), ("xof", c_int),
                ("generator_xy", ctypes.c_uint8 * 64), ("blinding_base_xy", ctypes.c_uint8 * 64), ("curve", c_int)]


class RingVerifierKeyStruct(ctypes.Structure):
    """dr_ring_verifier_key (include/dotring_hip.h)."""
    _fields_ = [("log2n", ctypes.c_uint), ("omega_n", ctypes.c_uint8 * 32), ("seed_xy", ctypes.c_uint8 * 64),
                ("fixed_commitments", ctypes.c_uint8 * 288), ("g1_generator", ctypes.c_uint8 * 96), ("g2", ctypes.c_uint8 * 384),
                ("fs_prefix", c_char_p), ("fs_prefix_len", c_size_t)]


_PROTOTYPES.update({
    "dr_ringvrf_verify_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), POINTER(RingVerifierKeyStruct), c_size_t, c_char_p, c_char_p,
                                        POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64),
                                        c_char_p, POINTER(c_int)]),
    "dr_g1_decompress_batch": (c_int, [c_void_p, c_char_p, c_size_t, c_char_p, c_char_p]),
    "dr_bsn_decode_points": (c_int, [c_void_p, c_char_p, c_size_t, c_char_p, c_char_p]),
    "dr_te_scalar_mul_batch": (c_int, [c_void_p, c_int, c_char_p, c_char_p, c_size_t, c_void_p]),
    "dr_te_msm": (c_int, [c_void_p, c_int, c_char_p, c_char_p, c_size_t, c_void_p]),
    "dr_te_msm_groups": (c_int, [c_void_p, c_int, c_char_p, c_char_p, c_size_t, c_size_t, c_void_p]),
    "dr_te_decode_points": (c_int, [c_void_p, c_int, c_char_p, c_size_t, c_char_p, c_char_p]),
    "dr_encode_to_curve_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                         POINTER(ctypes.c_uint64), c_size_t, c_char_p]),
    "dr_ring_prover_create_te": (c_int, [c_void_p, c_int, c_void_p, c_uint, c_uint, c_char_p, c_char_p, c_char_p, c_char_p, POINTER(c_void_p)]),
    "dr_pairing_selfcheck": (c_int, [c_char_p, c_char_p, c_size_t, POINTER(c_int)]),
    "dr_pedersen_prove_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_size_t, c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                        POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p, c_char_p, c_char_p]),
    "dr_pedersen_verify_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_size_t, c_char_p, c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                         POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), POINTER(c_int)]),
    "dr_ietf_prove_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_int, c_size_t, c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                    POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p, c_char_p, c_char_p]),
    "dr_ietf_verify_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_int, c_size_t, c_char_p, c_char_p, c_char_p, POINTER(ctypes.c_uint64),
                                     c_char_p, POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p]),
    "dr_ringvrf_prove_batch_multi": (c_int, [POINTER(c_void_p), c_size_t, POINTER(VrfSuiteStruct), c_size_t, c_char_p, POINTER(ctypes.c_uint64),
                                             c_char_p, POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                             POINTER(ctypes.c_uint32), c_char_p, c_size_t, c_char_p, c_char_p, c_char_p]),
    "dr_ringvrf_verify_batch_multi": (c_int, [POINTER(c_void_p), c_size_t, POINTER(VrfSuiteStruct), POINTER(RingVerifierKeyStruct), c_size_t,
                                              c_char_p, c_char_p, POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                              POINTER(ctypes.c_uint64), c_char_p, POINTER(c_int)]),
    "dr_host_hash": (c_int, [c_int, c_char_p, c_size_t, c_char_p, c_size_t]),
    "dr_hash_to_field_batch": (c_int, [POINTER(VrfSuiteStruct), c_char_p, POINTER(ctypes.c_uint64), c_size_t, c_char_p]),
    "dr_ringvrf_prove_batch": (c_int, [c_void_p, POINTER(VrfSuiteStruct), c_size_t, c_char_p, POINTER(ctypes.c_uint64), c_char_p,
                                       POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint64), c_char_p, POINTER(ctypes.c_uint32),
                                       c_char_p, c_size_t, c_char_p, c_char_p, c_char_p]),
})
_PROTOTYPES.update({
    "dr_te_fixed_base_msm_groups": (c_int, [c_void_p, c_int, c_char_p, c_size_t, c_char_p, c_size_t, c_void_p]),
    "dr_host_random_expand": (c_int, [c_char_p, c_void_p, c_size_t]),
    "dr_ringvrf_aux_take_blindings": (c_int, [c_void_p, c_size_t, c_void_p]),
    "dr_comm_unique_id": (c_int, [c_char_p]),
    "dr_comm_create": (c_int, [c_void_p, c_char_p, c_int, c_int, POINTER(c_void_p)]),
    "dr_comm_destroy": (None, [c_void_p]),
    "dr_comm_rank": (c_int, [c_void_p]),
    "dr_comm_world": (c_int, [c_void_p]),
    "dr_comm_count": (c_int, [c_void_p, POINTER(c_int)]),
    "dr_comm_all_gather": (c_int, [c_void_p, c_char_p, c_size_t, c_char_p]),
    "dr_g1_msm_sharded_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, POINTER(c_int)]),
})
COMM_ID_BYTES = 128
RINGVRF_AUX_BYTES = 960
PEDERSEN_AUX_BYTES = 288
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)


_buffers = threading.local()


def _thread_buffer(name: str, nbytes: int):
    """A per-thread ctypes byte buffer of at least nbytes, reused across calls (contents are overwritten by the next call)."""
    have = getattr(_buffers, name, None)
    if have is None or len(have) < nbytes:
        have = (ctypes.c_char * max(nbytes, 1))()
        setattr(_buffers, name, have)
    return have


def wipe(buf) -> None:
    """Zero a reused ctypes buffer that held secret material (the auxiliary records carry the blinding factors)."""
    ctypes.memset(buf, 0, len(buf))


def wipe_thread_buffer(name: str) -> None:
    """Zero the calling thread's reused buffer `name` if it exists (a native call that raised may have left secrets in it)."""
    have = getattr(_buffers, name, None)
    if have is not None:
        wipe(have)


def aux_take_blindings(aux_buf, batch: int):
    """The blinding factors of a dr_ringvrf_prove_batch call, moved out of its auxiliary records (zeroed there): returns the
    scratch buffer holding batch * 32 bytes; the caller slices each proof's own 32 bytes out and wipes it."""
    out = _thread_buffer("prove_blind", 32 * batch)
    _check(lib().dr_ringvrf_aux_take_blindings(aux_buf, batch, out))
    return out


def _ragged(items):
    """Concatenate byte strings -> (blob, uint64 offsets[count+1])."""
    off = array.array("Q", itertools.chain((0,), itertools.accumulate(map(len, items))))      # 0.05 ms per 1024 items (0.11 in a loop)
    return b"".join(items), (ctypes.c_uint64 * len(off)).from_buffer(off)


CURVE_BANDERSNATCH, CURVE_JUBJUB = 0, 1


def vrf_suite(suite_id: bytes, xof: bool, generator_xy: bytes, blinding_base_xy: bytes, curve: int = CURVE_BANDERSNATCH) -> VrfSuiteStruct:
    s = VrfSuiteStruct()
    s._keep = bytes(suite_id)
    s.suite_id, s.suite_id_len, s.xof, s.curve = s._keep, len(s._keep), 1 if xof else 0, curve
    ctypes.memmove(s.generator_xy, generator_xy, 64)
    ctypes.memmove(s.blinding_base_xy, blinding_base_xy, 64)
    return s


def ring_verifier_key(log2n: int, omega_n: int, seed_xy: bytes, fixed_commitments: bytes, g1_generator: bytes, g2: bytes,
                      fs_prefix: bytes) -> RingVerifierKeyStruct:
    vk = RingVerifierKeyStruct()
    vk.log2n = log2n
    ctypes.memmove(vk.omega_n, int(omega_n).to_bytes(32, "little"), 32)
    ctypes.memmove(vk.seed_xy, seed_xy, 64)
    ctypes.memmove(vk.fixed_commitments, fixed_commitments, 288)
    ctypes.memmove(vk.g1_generator, g1_generator, 96)
    ctypes.memmove(vk.g2, g2, 384)
    vk._keep = bytes(fs_prefix)
    vk.fs_prefix, vk.fs_prefix_len = vk._keep, len(vk._keep)
    return vk


def host_hash(kind: int, data: bytes, out_len: int) -> bytes:
    """kind: 0 SHA-512, 1 SHAKE128, 2 SHAKE256 (the library's own implementations; checked against hashlib in tests)."""
    out = ctypes.create_string_buffer(out_len)
    _check(lib().dr_host_hash(kind, data, len(data), out, out_len))
    return out.raw


def random_expand(seed32: bytes, nbytes: int) -> bytes:
    """nbytes of SHAKE256(seed || LE64(block)) output in 576-byte blocks, hashed on the library's worker threads: turns one
    32-byte secret seed from the OS into the hidden rows of a whole batch of ring proofs."""
    if len(seed32) != 32:
        raise ValueError("seed must be 32 bytes")
    out = ctypes.create_string_buffer(max(1, nbytes))
    _check(lib().dr_host_random_expand(seed32, out, nbytes))
    return out.raw[:nbytes]


def hash_to_field_batch(suite: VrfSuiteStruct, msgs) -> bytes:
    blob, off = _ragged(msgs)
    out = ctypes.create_string_buffer(max(1, 64 * len(msgs)))
    _check(lib().dr_hash_to_field_batch(byref(suite), blob, off, len(msgs), out))
    return out.raw[: 64 * len(msgs)]



def lib() -> ctypes.CDLL:
    """Load libdotring_hip.so (built by __graft_entry__.build() / `make -C dot_ring_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C dot_ring_amd/csrc` (hipcc, gfx950). "
                "dot_ring_amd has no CPU fallback."
            )
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(cdll, name)
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def last_error() -> str:
    return (lib().dr_last_error() or b"").decode("utf-8", "replace")


def _check(rc: int) -> None:
    if rc == DR_OK:
        return
    msg = (lib().dr_last_error() or b"").decode("utf-8", "replace")
    if rc in (DR_ERR_INVALID, DR_ERR_NOTSQUARE):
        raise ValueError(msg)
    if rc == DR_ERR_NOMEM:
        raise MemoryError(msg)
    raise DotRingHipError(msg)


def device_count() -> int:
    return int(lib().dr_device_count())


class DeviceBuffer:
    """A block of HBM owned by a Context."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = c_void_p()
        _check(lib().dr_dev_alloc(ctx.handle, nbytes, byref(self.ptr)))

    def upload(self, data: bytes) -> "DeviceBuffer":
        if len(data) > self.nbytes:
            raise ValueError("upload larger than the buffer")
        _check(lib().dr_dev_upload(self.ctx.handle, self.ptr, data, len(data)))
        return self

    def download(self, nbytes: int | None = None) -> bytes:
        nbytes = self.nbytes if nbytes is None else nbytes
        out = ctypes.create_string_buffer(nbytes)
        _check(lib().dr_dev_download(self.ctx.handle, out, self.ptr, nbytes))
        return out.raw

    def free(self) -> None:
        if self.ptr:
            lib().dr_dev_free(self.ctx.handle, self.ptr)
            self.ptr = c_void_p()


class Srs:
    """SRS bases resident in HBM (Montgomery-form affine)."""

    def __init__(self, ctx: "Context", g1_be_xy: bytes | None = None, *, synthetic_seed: bytes | None = None, first: int = 1, count: int = 0,
                 tau: int | None = None):
        self.ctx = ctx
        self.handle = c_void_p()
        if tau is not None:
            self.count = count
            _check(lib().dr_srs_powers(ctx.handle, synthetic_seed, int(tau).to_bytes(32, "little"), count, byref(self.handle)))
            return
        if synthetic_seed is not None:
            self.count = count
            _check(lib().dr_srs_synthetic(ctx.handle, synthetic_seed, first, count, byref(self.handle)))
            return
        if g1_be_xy is None or len(g1_be_xy) % 96:
            raise ValueError("SRS bytes must be a multiple of 96")
        self.count = len(g1_be_xy) // 96
        _check(lib().dr_srs_load(ctx.handle, g1_be_xy, self.count, byref(self.handle)))

    def precompute(self, window_bits: int) -> "Srs":
        """Build (or with 0 drop) the fixed-base window table in HBM."""
        _check(lib().dr_srs_precompute(self.ctx.handle, self.handle, window_bits))
        return self

    def table_info(self, n: int = 0, batch: int = 0) -> dict:
        """Shape of the fixed-base table and the tiling `batch` MSMs of n points over it take (dr_srs_table_info): window_bits, rows,
        digit rows per scalar, the per-call tiling (window rows or the non-adjacent form over a bit-row table) with its width, and
        the expected non-zero digits per scalar (= bucket additions per pair)."""
        info = (c_int * 6)()
        _check(lib().dr_srs_table_info(self.handle, n, batch, info))
        return {"window_bits": info[0], "rows": info[1], "batched_windows": info[2], "tiling_bits": info[3],
                "tiling": ("window rows", "-", "non-adjacent form")[info[4]], "digits_per_scalar": info[5] / 1000.0,
                "odd_buckets": bool(info[4])}

    def download(self, offset: int, count: int) -> bytes:
        out = ctypes.create_string_buffer(max(96 * count, 1))
        _check(lib().dr_srs_download(self.ctx.handle, self.handle, offset, count, out))
        return out.raw[: 96 * count]

    def close(self) -> None:
        if self.handle:
            lib().dr_srs_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingProver:
    """Device-resident batched ring prover for one ring (dr_ring_prover_*)."""

    def __init__(self, ctx: "Context", srs: Srs, log2n: int, max_ring: int, omega_n: int, omega_4n: int, nm_points_xy: bytes, seed_xy: bytes,
                 curve: int = CURVE_BANDERSNATCH):
        self.ctx, self.srs = ctx, srs
        self.n = 1 << log2n
        self.handle = c_void_p()
        _check(lib().dr_ring_prover_create_te(ctx.handle, curve, srs.handle, log2n, max_ring, omega_n.to_bytes(32, "little"),
                                              omega_4n.to_bytes(32, "little"), nm_points_xy, seed_xy, byref(self.handle)))

    def close(self) -> None:
        if self.handle:
            lib().dr_ring_prover_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        # provers are cached on Ring objects: when the ring goes, its tables and per-batch state (hundreds of MB of HBM)
        # must go too
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _points(raw: bytes, inf, count: int) -> list:
        return [None if inf[i] else raw[96 * i : 96 * i + 96] for i in range(count)]

    def root(self) -> list:
        out, inf = ctypes.create_string_buffer(3 * 96), (c_int * 3)()
        _check(lib().dr_ring_prover_root(self.handle, out, inf))
        return self._points(out.raw, inf, 3)

    def fixed_coeffs(self) -> bytes:
        out = ctypes.create_string_buffer(3 * self.n * 32)
        _check(lib().dr_ring_prover_fixed_coeffs(self.handle, out))
        return out.raw

    def witness(self, producer_index: list, blinding: bytes, zk_rows: bytes | None):
        batch = len(producer_index)
        idx = (ctypes.c_uint32 * batch)(*producer_index)
        rel, cms, inf = ctypes.create_string_buffer(64 * batch), ctypes.create_string_buffer(4 * 96 * batch), (c_int * (4 * batch))()
        _check(lib().dr_ring_prove_witness(self.handle, batch, idx, blinding, zk_rows, rel, cms, inf))
        return rel.raw, self._points(cms.raw, inf, 4 * batch)

    def quotient(self, batch: int, alphas: bytes) -> list:
        out, inf = ctypes.create_string_buffer(96 * batch), (c_int * batch)()
        _check(lib().dr_ring_prove_quotient(self.handle, batch, alphas, out, inf))
        return self._points(out.raw, inf, batch)

    def evals(self, batch: int, zetas: bytes) -> bytes:
        out = ctypes.create_string_buffer(8 * 32 * batch)
        _check(lib().dr_ring_prove_evals(self.handle, batch, zetas, out))
        return out.raw

    def openings(self, batch: int, nus: bytes) -> list:
        out, inf = ctypes.create_string_buffer(2 * 96 * batch), (c_int * (2 * batch))()
        _check(lib().dr_ring_prove_openings(self.handle, batch, nus, out, inf))
        return self._points(out.raw, inf, 2 * batch)

    def wipe(self) -> None:
        """Zero the per-batch state and the MSM scratch on the device now (every batch already ends with it)."""
        _check(lib().dr_ring_prover_wipe(self.handle))

    def residue(self) -> int:
        """Non-zero 32-bit words left in the prover's per-batch device state and its contexts' scratch (0 after a wipe)."""
        words = ctypes.c_uint64(0)
        _check(lib().dr_ring_prover_residue(self.handle, byref(words)))
        return words.value

    def ringvrf_prove_batch(self, suite: "VrfSuiteStruct", alphas, ads, salts, secret_scalars: bytes, producer_index: list,
                            fs_prefix: bytes, zk_random48: bytes | None):
        """dr_ringvrf_prove_batch: the whole batch (hashing on the library's worker threads, GPU phases in between).
        Returns (batch * 784 proof bytes, batch * 960 auxiliary bytes)."""
        batch = len(alphas)
        a_blob, a_off = _ragged(alphas)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        idx = (ctypes.c_uint32 * batch)(*producer_index)
        # output buffers are kept per thread and reused: two fresh megabyte-sized buffers plus their `.raw` copies cost ~1 ms per
        # 1024 proofs in page faults.  The caller slices its proofs out (slicing a c_char array yields bytes) before its next call.
        out, aux = _thread_buffer("prove_out", 784 * batch), _thread_buffer("prove_aux", RINGVRF_AUX_BYTES * batch)
        _check(lib().dr_ringvrf_prove_batch(self.handle, byref(suite), batch, a_blob, a_off, d_blob, d_off, s_blob, s_off, secret_scalars, idx,
                                            fs_prefix, len(fs_prefix), zk_random48, out, aux))
        return out, aux


def ringvrf_prove_batch_multi(provers: list, suite: "VrfSuiteStruct", alphas, ads, salts, secret_scalars: bytes, producer_index: list,
                              fs_prefix: bytes, zk_random48: bytes | None):
    """dr_ringvrf_prove_batch_multi: ONE batch over the devices of `provers` (a RingProver of the same ring per device), device g
    proving proofs [g B / G, (g + 1) B / G).  Returns the same (proof bytes, auxiliary bytes) buffers as RingProver.ringvrf_prove_batch."""
    batch = len(alphas)
    a_blob, a_off = _ragged(alphas)
    d_blob, d_off = _ragged(ads)
    s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
    idx = (ctypes.c_uint32 * batch)(*producer_index)
    handles = (c_void_p * len(provers))(*[p.handle for p in provers])
    out, aux = _thread_buffer("prove_out", 784 * batch), _thread_buffer("prove_aux", RINGVRF_AUX_BYTES * batch)
    _check(lib().dr_ringvrf_prove_batch_multi(handles, len(provers), byref(suite), batch, a_blob, a_off, d_blob, d_off, s_blob, s_off,
                                              secret_scalars, idx, fs_prefix, len(fs_prefix), zk_random48, out, aux))
    return out, aux


def ringvrf_verify_batch_multi(contexts: list, suite: "VrfSuiteStruct", vk: "RingVerifierKeyStruct", proofs: bytes, inputs, ads, salts,
                               seed32: bytes) -> bool:
    """dr_ringvrf_verify_batch_multi: ONE batch of 784-byte proofs over the devices of `contexts`, the shards' verdicts AND-ed."""
    batch = len(inputs)
    if len(proofs) != 784 * batch:
        raise ValueError("proofs must be 784 bytes each")
    i_blob, i_off = _ragged(inputs)
    d_blob, d_off = _ragged(ads)
    s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
    handles = (c_void_p * len(contexts))(*[c.handle for c in contexts])
    ok = c_int(0)
    _check(lib().dr_ringvrf_verify_batch_multi(handles, len(contexts), byref(suite), byref(vk), batch, proofs, i_blob, i_off, d_blob, d_off,
                                               s_blob, s_off, seed32, byref(ok)))
    return bool(ok.value)


class Context:
    """One GPU + one HIP stream.  Not thread-safe; create one per thread / per rank."""

    def __init__(self, device_id: int = 0):
        self.handle = c_void_p()
        _check(lib().dr_ctx_create(device_id, byref(self.handle)))
        self.device_id = device_id

    def close(self) -> None:
        if self.handle:
            lib().dr_ctx_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        _check(lib().dr_ctx_sync(self.handle))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def scratch_residue(self) -> int:
        """Non-zero 32-bit words in this context's scratch buffers (0 after a batch prover has wiped them)."""
        words = ctypes.c_uint64(0)
        _check(lib().dr_ctx_scratch_residue(self.handle, byref(words)))
        return words.value

    # ---- profiling
    def prof_enable(self, on: bool = True) -> None:
        _check(lib().dr_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self) -> None:
        _check(lib().dr_prof_reset(self.handle))

    def prof_get(self, kernel: str) -> tuple[float, int]:
        ms, cnt = c_double(0), c_int(0)
        _check(lib().dr_prof_get(self.handle, kernel.encode(), byref(ms), byref(cnt)))
        return ms.value, cnt.value

    def fr_ops_selftest(self, a: bytes, b: bytes):
        """dr_fr_ops_selftest: (n x 12 x 32 result bytes, n square flags) for n pairs of canonical 32-byte elements."""
        n = len(a) // 32
        if len(a) != 32 * n or len(b) != 32 * n:
            raise ValueError("operands are 32 bytes each")
        out, flags = ctypes.create_string_buffer(max(1, 384 * n)), ctypes.create_string_buffer(max(1, n))
        _check(lib().dr_fr_ops_selftest(self.handle, a, b, n, out, flags))
        return out.raw[: 384 * n], flags.raw[:n]

    # ---- seam A
    # (curve = CURVE_BANDERSNATCH / CURVE_JUBJUB; the default goes through the dr_bsn_* names of the original seam)
    def bsn_scalar_mul_batch(self, pts_xy: bytes, scalars: bytes, curve: int = CURVE_BANDERSNATCH) -> bytes:
        n = len(scalars) // 32
        if len(scalars) != 32 * n or len(pts_xy) != 64 * n:
            raise ValueError("Points and scalars must have same length")
        out = ctypes.create_string_buffer(max(64 * n, 1))
        if curve == CURVE_BANDERSNATCH:
            _check(lib().dr_bsn_scalar_mul_batch(self.handle, pts_xy, scalars, n, out))
        else:
            _check(lib().dr_te_scalar_mul_batch(self.handle, curve, pts_xy, scalars, n, out))
        return out.raw[: 64 * n]

    def bsn_scalar_mul_batch_dev(self, d_pts: DeviceBuffer, d_scalars: DeviceBuffer, n: int, d_out: DeviceBuffer) -> None:
        _check(lib().dr_bsn_scalar_mul_batch_dev(self.handle, d_pts.ptr, d_scalars.ptr, n, d_out.ptr))

    def bsn_msm(self, pts_xy: bytes, scalars: bytes, curve: int = CURVE_BANDERSNATCH) -> bytes:
        n = len(scalars) // 32
        if len(scalars) != 32 * n or len(pts_xy) != 64 * n:
            raise ValueError("Points and scalars must have same length")
        out = ctypes.create_string_buffer(64)
        if curve == CURVE_BANDERSNATCH:
            _check(lib().dr_bsn_msm(self.handle, pts_xy, scalars, n, out))
        else:
            _check(lib().dr_te_msm(self.handle, curve, pts_xy, scalars, n, out))
        return out.raw

    def bsn_msm_groups(self, pts_xy: bytes, scalars: bytes, m: int, curve: int = CURVE_BANDERSNATCH) -> bytes:
        n = len(scalars) // 32
        if m <= 0 or n % m or len(scalars) != 32 * n or len(pts_xy) != 64 * n:
            raise ValueError("Points and scalars must have same length (a multiple of the group size)")
        groups = n // m
        out = ctypes.create_string_buffer(max(64 * groups, 1))
        if curve == CURVE_BANDERSNATCH:
            _check(lib().dr_bsn_msm_groups(self.handle, pts_xy, scalars, groups, m, out))
        else:
            _check(lib().dr_te_msm_groups(self.handle, curve, pts_xy, scalars, groups, m, out))
        return out.raw[: 64 * groups]

    def te_fixed_base_msm_groups(self, bases_xy: bytes, scalars: bytes, curve: int = CURVE_BANDERSNATCH) -> bytes:
        """out[g] = sum_j scalars[g*m+j] * bases[j] over m = len(bases_xy) / 64 constant bases (fixed-base window tables)."""
        m = len(bases_xy) // 64
        if m == 0 or len(bases_xy) != 64 * m or len(scalars) % (32 * m):
            raise ValueError("Points and scalars must have same length (a multiple of the number of bases)")
        groups = len(scalars) // (32 * m)
        out = ctypes.create_string_buffer(max(64 * groups, 1))
        _check(lib().dr_te_fixed_base_msm_groups(self.handle, curve, bases_xy, m, scalars, groups, out))
        return out.raw[: 64 * groups]

    def encode_to_curve_batch(self, suite: "VrfSuiteStruct", msgs, salts=None) -> bytes:
        """dr_encode_to_curve_batch: encode_to_curve(salt_i || msg_i) by the suite's own method -> count * 64 bytes x||y."""
        count = len(msgs)
        m_blob, m_off = _ragged([bytes(m) for m in msgs])
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged([bytes(x) for x in salts])
        out = ctypes.create_string_buffer(max(64 * count, 1))
        _check(lib().dr_encode_to_curve_batch(self.handle, byref(suite), m_blob, m_off, s_blob, s_off, count, out))
        return out.raw[: 64 * count]

    def bsn_encode_to_curve_batch(self, u_pairs: bytes) -> bytes:
        n = len(u_pairs) // 64
        if len(u_pairs) != 64 * n:
            raise ValueError("u_pairs must be n * 64 bytes")
        out = ctypes.create_string_buffer(max(64 * n, 1))
        _check(lib().dr_bsn_encode_to_curve_batch(self.handle, u_pairs, n, out))
        return out.raw[: 64 * n]

    # ---- seam B
    def srs_load(self, g1_be_xy: bytes) -> Srs:
        return Srs(self, g1_be_xy)

    def srs_synthetic(self, seed_be_xy: bytes, count: int, first: int = 1) -> Srs:
        """bases[i] = (first+i) * seed, generated on the GPU."""
        return Srs(self, synthetic_seed=seed_be_xy, first=first, count=count)

    def ringvrf_verify_batch(self, suite: "VrfSuiteStruct", vk: "RingVerifierKeyStruct", proofs: bytes, inputs, ads, salts, seed32: bytes) -> bool:
        """dr_ringvrf_verify_batch over 784-byte encoded proofs (<= 4096 per call)."""
        batch = len(inputs)
        if len(proofs) != 784 * batch:
            raise ValueError("proofs must be 784 bytes each")
        i_blob, i_off = _ragged(inputs)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        ok = c_int(0)
        _check(lib().dr_ringvrf_verify_batch(self.handle, byref(suite), byref(vk), batch, proofs, i_blob, i_off, d_blob, d_off, s_blob, s_off,
                                             seed32, byref(ok)))
        return bool(ok.value)

    def g1_decompress_batch(self, enc: bytes):
        """KZG.decompress_g1 for len(enc)/48 encodings on the GPU -> (list of 96-byte records or None for infinity, flags)."""
        if len(enc) % 48:
            raise ValueError("compressed G1 points are 48 bytes each")
        count = len(enc) // 48
        out, ok = ctypes.create_string_buffer(max(1, 96 * count)), ctypes.create_string_buffer(max(1, count))
        _check(lib().dr_g1_decompress_batch(self.handle, enc, count, out, ok))
        raw = out.raw
        zero = bytes(96)
        pts = [None if raw[96 * i : 96 * i + 96] == zero else raw[96 * i : 96 * i + 96] for i in range(count)]
        return pts, ok.raw[:count]

    def bsn_decode_points(self, enc: bytes, curve: int = CURVE_BANDERSNATCH):
        """dec_point for len(enc)/32 compressed points on the GPU -> (affine x||y bytes, validity flags)."""
        if len(enc) % 32:
            raise ValueError("compressed points are 32 bytes each")
        count = len(enc) // 32
        out, ok = ctypes.create_string_buffer(max(1, 64 * count)), ctypes.create_string_buffer(max(1, count))
        if curve == CURVE_BANDERSNATCH:
            _check(lib().dr_bsn_decode_points(self.handle, enc, count, out, ok))
        else:
            _check(lib().dr_te_decode_points(self.handle, curve, enc, count, out, ok))
        return out.raw[: 64 * count], ok.raw[:count]

    def pedersen_prove_batch(self, suite: "VrfSuiteStruct", alphas, ads, salts, secret_scalars: bytes):
        """dr_pedersen_prove_batch -> (batch * 192 proof bytes, batch * 288 auxiliary bytes)."""
        batch = len(alphas)
        a_blob, a_off = _ragged(alphas)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        out, aux = ctypes.create_string_buffer(max(1, 192 * batch)), ctypes.create_string_buffer(max(1, PEDERSEN_AUX_BYTES * batch))
        _check(lib().dr_pedersen_prove_batch(self.handle, byref(suite), batch, a_blob, a_off, d_blob, d_off, s_blob, s_off, secret_scalars, out, aux))
        return out.raw[: 192 * batch], aux.raw[: PEDERSEN_AUX_BYTES * batch]

    def ietf_prove_batch(self, suite: "VrfSuiteStruct", thin: bool, alphas, ads, salts, secret_scalars: bytes):
        """dr_ietf_prove_batch -> (batch * (96 if thin else 80) proof bytes, batch * 128 bytes: O and R affine)."""
        batch, plen = len(alphas), 96 if thin else 80
        a_blob, a_off = _ragged(alphas)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        out, aux = ctypes.create_string_buffer(max(1, plen * batch)), ctypes.create_string_buffer(max(1, 128 * batch))
        _check(lib().dr_ietf_prove_batch(self.handle, byref(suite), 1 if thin else 0, batch, a_blob, a_off, d_blob, d_off, s_blob, s_off,
                                         secret_scalars, out, aux))
        return out.raw[: plen * batch], aux.raw[: 128 * batch]

    def ietf_verify_batch(self, suite: "VrfSuiteStruct", thin: bool, proofs: bytes, public_keys: bytes, inputs, ads, salts) -> bytes:
        """dr_ietf_verify_batch over encoded Tiny (80-byte) / Thin (96-byte) proofs: one verdict byte per proof — 1 verifies, 0 does
        not, 2 invalid public key, 3 malformed proof."""
        batch, plen = len(inputs), 96 if thin else 80
        if len(proofs) != plen * batch or len(public_keys) != 32 * batch:
            raise ValueError(f"proofs must be {plen} bytes and public keys 32 bytes each")
        i_blob, i_off = _ragged(inputs)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        verdict = ctypes.create_string_buffer(max(1, batch))
        _check(lib().dr_ietf_verify_batch(self.handle, byref(suite), 1 if thin else 0, batch, proofs, public_keys, i_blob, i_off, d_blob, d_off,
                                          s_blob, s_off, verdict))
        return verdict.raw[:batch]

    def pedersen_verify_batch(self, suite: "VrfSuiteStruct", proofs: bytes, inputs, ads, salts) -> bool:
        batch = len(inputs)
        if len(proofs) != 192 * batch:
            raise ValueError("proofs must be 192 bytes each")
        i_blob, i_off = _ragged(inputs)
        d_blob, d_off = _ragged(ads)
        s_blob, s_off = (None, None) if not salts or not any(salts) else _ragged(salts)
        ok = c_int(0)
        _check(lib().dr_pedersen_verify_batch(self.handle, byref(suite), batch, proofs, i_blob, i_off, d_blob, d_off, s_blob, s_off, byref(ok)))
        return bool(ok.value)

    def srs_powers(self, base_be_xy: bytes, tau: int, count: int) -> Srs:
        """bases[i] = tau^i * base (known-tau SRS for tests/benchmarks beyond the shipped file), generated on the GPU."""
        return Srs(self, synthetic_seed=base_be_xy, tau=tau, count=count)

    def g1_msm(self, srs: Srs, scalars: bytes, offset: int = 0) -> bytes | None:
        """Affine BE x||y (96 bytes) or None for the point at infinity."""
        n = len(scalars) // 32
        if len(scalars) != 32 * n:
            raise ValueError("scalars must be a multiple of 32 bytes")
        out, inf = ctypes.create_string_buffer(96), c_int(0)
        _check(lib().dr_g1_msm(self.handle, srs.handle, offset, scalars, n, out, byref(inf)))
        return None if inf.value else out.raw

    def g1_msm_dev(self, srs: Srs, d_scalars: DeviceBuffer, n: int, offset: int = 0) -> bytes | None:
        out, inf = ctypes.create_string_buffer(96), c_int(0)
        _check(lib().dr_g1_msm_dev(self.handle, srs.handle, offset, d_scalars.ptr, n, out, byref(inf)))
        return None if inf.value else out.raw

    def g1_msm_batch(self, srs: Srs, scalars: bytes, n: int) -> list[bytes | None]:
        if n <= 0 or len(scalars) % (32 * n):
            raise ValueError("scalars must be batch * n * 32 bytes")
        batch = len(scalars) // (32 * n)
        out, inf = ctypes.create_string_buffer(96 * batch), (c_int * batch)()
        _check(lib().dr_g1_msm_batch(self.handle, srs.handle, scalars, n, batch, out, inf))
        return [None if inf[b] else out.raw[96 * b : 96 * b + 96] for b in range(batch)]

    def g1_msm_batch_dev(self, srs: Srs, d_scalars: DeviceBuffer, n: int, batch: int) -> list[bytes | None]:
        out, inf = ctypes.create_string_buffer(96 * batch), (c_int * batch)()
        _check(lib().dr_g1_msm_batch_dev(self.handle, srs.handle, d_scalars.ptr, n, batch, out, inf))
        return [None if inf[b] else out.raw[96 * b : 96 * b + 96] for b in range(batch)]

    def g1_msm_points(self, pts_be_xy: bytes, scalars: bytes) -> bytes | None:
        n = len(scalars) // 32
        if len(scalars) != 32 * n or len(pts_be_xy) != 96 * n:
            raise ValueError("Points and scalars must have same length")
        out, inf = ctypes.create_string_buffer(96), c_int(0)
        _check(lib().dr_g1_msm_points(self.handle, pts_be_xy, scalars, n, out, byref(inf)))
        return None if inf.value else out.raw

    # ---- seam C
    def ntt(self, data: bytes, log2n: int, omega: int, scale: int | None = None) -> bytes:
        n = 1 << log2n
        if len(data) % (32 * n):
            raise ValueError(f"coefficient length does not match native NTT plan size {n}")
        batch = len(data) // (32 * n)
        buf = ctypes.create_string_buffer(data, len(data))
        sc = scale.to_bytes(32, "little") if scale is not None else None
        _check(lib().dr_ntt(self.handle, buf, log2n, batch, omega.to_bytes(32, "little"), sc))
        return buf.raw

    def ntt_dev(self, d_data: DeviceBuffer, log2n: int, batch: int, omega: int, scale: int | None = None) -> None:
        sc = scale.to_bytes(32, "little") if scale is not None else None
        _check(lib().dr_ntt_dev(self.handle, d_data.ptr, log2n, batch, omega.to_bytes(32, "little"), sc))


# ---- host-only helpers (no GPU needed)
def fr_sqrt(v: int) -> int:
    out = ctypes.create_string_buffer(32)
    _check(lib().dr_fr_sqrt(int(v).to_bytes(32, "little"), out))
    return int.from_bytes(out.raw, "little")


def g1_sum(points: list) -> bytes | None:
    """Host-side sum of a few affine points (96-byte BE records or None)."""
    raw = b"".join(bytes(96) if p is None else p for p in points)
    out, inf = ctypes.create_string_buffer(96), c_int(0)
    _check(lib().dr_g1_sum(raw, len(points), out, byref(inf)))
    return None if inf.value else out.raw


def g2_mul(g2_be: bytes, scalar: int) -> bytes:
    """scalar * Q for a 192-byte zcash-layout G2 point (host; setup of known-tau test SRS only)."""
    out = ctypes.create_string_buffer(192)
    _check(lib().dr_g2_mul(g2_be, int(scalar).to_bytes(32, "little"), out))
    return out.raw


def pairing_check(pairs: list) -> bool:
    """True iff prod e(P_i, Q_i) == 1; pairs = [(g1_96_bytes_or_None, g2_192_bytes), ...]. Host-side."""
    g1 = b"".join(bytes(96) if p is None else p for p, _ in pairs)
    g2 = b"".join(q for _, q in pairs)
    ok = c_int(0)
    _check(lib().dr_pairing_check(g1, g2, len(pairs), byref(ok)))
    return bool(ok.value)


def g1_neg(xy: bytes | None) -> bytes | None:
    if xy is None:
        return None
    y = int.from_bytes(xy[48:], "big")
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    return xy[:48] + ((p - y) % p).to_bytes(48, "big")


def g1_compress(xy: bytes | None) -> bytes:
    out = ctypes.create_string_buffer(48)
    _check(lib().dr_g1_compress(xy if xy is not None else bytes(96), 1 if xy is None else 0, out))
    return out.raw


def g1_decompress(data: bytes) -> bytes | None:
    if len(data) != 48:
        raise ValueError(f"invalid BLS12-381 G1 length: expected 48, got {len(data)}")
    out, inf = ctypes.create_string_buffer(96), c_int(0)
    _check(lib().dr_g1_decompress(data, out, byref(inf)))
    return None if inf.value else out.raw
