"""ctypes binding of libzkmi355.so (the C ABI in include/zkmi355.h) for tests and bench.py.

This is test/bench plumbing over the product library: the reference's host language (Rust) is not
available in this image, so the production host side is the C++ layer in ``csrc/``; see
INTEGRATION.md for the Rust ``extern "C"`` block a maintainer would add.

There is NO CPU fallback: if the shared library or a gfx950 device is missing, loading /
context creation raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzkmi355.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "zkmi355.h")

FIELD_FR, FIELD_FQ = 0, 1
OP_ADD, OP_SUB, OP_MUL = 0, 1, 2
# quotient-program opcodes (csrc/quotient.hip)
Q_END, Q_PUSH_COL, Q_PUSH_CONST, Q_ADD, Q_SUB, Q_MUL, Q_NEG, Q_SQUARE, Q_DOUBLE, Q_FOLD, Q_MUL_CONST, Q_ADD_CONST, Q_TEE_TMP, Q_PUSH_TMP = range(14)

_lib = None


class ZkError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into lib/libzkmi355.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _HERE, "-j8", "-s"]
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("ZKMI355_LIB", LIB_PATH)       # A/B measurements against another build of the library
        if not os.path.exists(path):
            raise ZkError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        _lib = ctypes.CDLL(path)
        _lib.zk_last_error.restype = ctypes.c_char_p
        _lib.zk_version.restype = ctypes.c_char_p
        _lib.zk_srs_g.restype = ctypes.c_void_p
        _lib.zk_srs_g_lagrange.restype = ctypes.c_void_p
        _lib.zk_srs_k.restype = ctypes.c_uint32
        _lib.zk_params_file_len.restype = ctypes.c_size_t
        _lib.zk_ctx_destroy.restype = None
        _lib.zk_srs_destroy.restype = None
        _lib.zk_pk_destroy.restype = None
        _lib.zk_proof_abort.restype = None
        _lib.zk_transcript_new.restype = ctypes.c_void_p
        _lib.zk_transcript_free.restype = None
        _lib.zk_transcript_free.argtypes = [ctypes.c_void_p]
        _lib.zk_transcript_proof.restype = ctypes.c_size_t
    return _lib


def _host_ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


class DeviceBuffer:
    """Owning handle of a device allocation made through the C ABI."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        ctx._ck(lib().zk_buf_alloc(ctx.h, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().zk_buf_free(self.ctx.h, ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        self.ctx._ck(lib().zk_h2d(self.ctx.h, ctypes.c_void_p(self.ptr), _host_ptr(a), ctypes.c_size_t(a.nbytes)))
        return self

    def download(self, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._ck(lib().zk_d2h(self.ctx.h, _host_ptr(out), ctypes.c_void_p(self.ptr), ctypes.c_size_t(out.nbytes)))
        return out


class Srs:
    def __init__(self, ctx: "Context", handle):
        self.ctx, self.h = ctx, handle

    @property
    def k(self) -> int:
        return lib().zk_srs_k(self.h)

    @property
    def g_ptr(self) -> int:
        return lib().zk_srs_g(self.h)

    @property
    def g_lagrange_ptr(self) -> Optional[int]:
        return lib().zk_srs_g_lagrange(self.h)

    def downsize(self, new_k: int) -> "Srs":
        """ParamsKZG::downsize: first 2^new_k points of g, Lagrange basis recomputed on the device."""
        h = ctypes.c_void_p()
        self.ctx._ck(lib().zk_srs_downsize(self.ctx.h, self.h, ctypes.c_uint32(new_k), ctypes.byref(h)))
        return Srs(self.ctx, h)

    def download_g(self) -> np.ndarray:
        n = 1 << self.k
        out = np.empty((n, 8), dtype=np.uint64)
        self.ctx._ck(lib().zk_d2h(self.ctx.h, _host_ptr(out), ctypes.c_void_p(self.g_ptr), ctypes.c_size_t(out.nbytes)))
        return out

    def download_g_lagrange(self) -> np.ndarray:
        n = 1 << self.k
        out = np.empty((n, 8), dtype=np.uint64)
        self.ctx._ck(lib().zk_d2h(self.ctx.h, _host_ptr(out), ctypes.c_void_p(self.g_lagrange_ptr), ctypes.c_size_t(out.nbytes)))
        return out

    def destroy(self):
        if self.h:
            lib().zk_srs_destroy(self.ctx.h, self.h)
            self.h = None


class ProvingKey:
    def __init__(self, ctx: "Context", handle):
        self.ctx, self.h = ctx, handle

    def vk(self, num_commitments: int):
        """(commitments (num, 8) u64 affine Montgomery, vk_repr (4,) u64 Montgomery)."""
        com = np.zeros((max(num_commitments, 1), 8), dtype=np.uint64)
        rep = np.zeros(4, dtype=np.uint64)
        self.ctx._ck(lib().zk_pk_vk(self.ctx.h, self.h, _host_ptr(com), _host_ptr(rep)))
        return com[:num_commitments], rep

    def set_transcript_repr(self, repr_mont: np.ndarray):
        """install halo2's `vk.transcript_repr()` ((4,) u64 Montgomery Fr): the first scalar every proof absorbs"""
        r = np.ascontiguousarray(repr_mont, dtype=np.uint64).reshape(4)
        self.ctx._ck(lib().zk_pk_set_transcript_repr(self.ctx.h, self.h, _host_ptr(r)))

    def shape(self) -> dict:
        out = (ctypes.c_uint32 * 16)()
        self.ctx._ck(lib().zk_pk_shape(self.ctx.h, self.h, out))
        names = ["k", "degree", "extended_k", "F", "A", "I", "P", "C", "L", "phases", "challenges", "blinding_factors",
                 "advice_queries", "fixed_queries", "commitments", "evaluations"]
        return dict(zip(names, list(out)))

    def quotient_plan(self) -> dict:
        """how the key's constraints are dealt to degree classes and what each class's compiled program costs (zk_pk_quotient_plan)"""
        out = (ctypes.c_uint32 * (8 + 8 * 16))()
        self.ctx._ck(lib().zk_pk_quotient_plan(self.ctx.h, self.h, out, ctypes.c_size_t(len(out))))
        return plan_summary(list(out))

    def destroy(self):
        if self.h:
            lib().zk_pk_destroy(self.ctx.h, self.h)
            self.h = None


def plan_summary(words) -> dict:
    """the summary words of zk_host_quotient_plan / zk_pk_quotient_plan as a dict"""
    E = int(words[0])
    names = ("used", "instructions", "products", "columns", "values_parked", "slots_alive", "factor_groups", "last")
    return {"classes_E": E, "constraints": int(words[1]), "degree_classes": int(words[2]), "additive_split": int(words[3]), "expression_graph": int(words[4]),
            "remainders": int(words[5]), "cost_estimate": int(words[6]), "columns": int(words[7]),
            "classes": [dict(zip(names, (int(v) for v in words[8 + 8 * e:16 + 8 * e]))) for e in range(E + 1)]}


TRANSCRIPT_BLAKE2B, TRANSCRIPT_POSEIDON, TRANSCRIPT_EVM = 0, 1, 2


class HostTranscript:
    """zk_transcript: the library's Blake2b / Poseidon / EVM transcript as a host-only object (no GPU)."""

    def __init__(self, kind: int):
        self.h = lib().zk_transcript_new(ctypes.c_int(kind))
        if not self.h:
            raise ZkError(f"unknown transcript kind {kind}")

    def _ck(self, rc):
        if rc != 0:
            raise ZkError(f"transcript operation failed with status {rc}")

    def common_point(self, affine64: bytes): self._ck(lib().zk_transcript_common_point(ctypes.c_void_p(self.h), ctypes.c_char_p(bytes(affine64))))
    def common_scalar(self, fr32: bytes): self._ck(lib().zk_transcript_common_scalar(ctypes.c_void_p(self.h), ctypes.c_char_p(bytes(fr32))))
    def write_point(self, affine64: bytes): self._ck(lib().zk_transcript_write_point(ctypes.c_void_p(self.h), ctypes.c_char_p(bytes(affine64))))
    def write_scalar(self, fr32: bytes): self._ck(lib().zk_transcript_write_scalar(ctypes.c_void_p(self.h), ctypes.c_char_p(bytes(fr32))))

    def squeeze_challenge(self) -> bytes:
        out = ctypes.create_string_buffer(32)
        self._ck(lib().zk_transcript_squeeze(ctypes.c_void_p(self.h), out))
        return out.raw

    def proof(self) -> bytes:
        p = ctypes.c_void_p()
        n = lib().zk_transcript_proof(ctypes.c_void_p(self.h), ctypes.byref(p))
        return ctypes.string_at(p.value, n) if n else b""

    def close(self):
        if self.h:
            lib().zk_transcript_free(ctypes.c_void_p(self.h))
            self.h = None


def host_keccak256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    rc = lib().zk_host_keccak256(ctypes.c_char_p(data), ctypes.c_size_t(len(data)), out)
    if rc != 0:
        raise ZkError(f"zk_host_keccak256 failed with status {rc}")
    return out.raw


def host_poseidon_permute(state_mont: np.ndarray) -> np.ndarray:
    """(5, 4) u64 Montgomery Fr in and out: one permutation of the transcript's Poseidon (T 5, R_F 8, R_P 60)"""
    st = np.ascontiguousarray(state_mont, dtype=np.uint64).reshape(5, 4).copy()
    rc = lib().zk_host_poseidon_permute(_host_ptr(st))
    if rc != 0:
        raise ZkError(f"zk_host_poseidon_permute failed with status {rc}")
    return st


def host_poseidon_permute_width3(state_mont: np.ndarray) -> np.ndarray:
    """(3, 4) u64 Montgomery Fr in and out: the same generator at T 3, R_F 8, R_P 57 (the reference's Poseidon code hash width)"""
    st = np.ascontiguousarray(state_mont, dtype=np.uint64).reshape(3, 4).copy()
    rc = lib().zk_host_poseidon_permute_width3(_host_ptr(st))
    if rc != 0:
        raise ZkError(f"zk_host_poseidon_permute_width3 failed with status {rc}")
    return st


_TR_IN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
_TR_OUT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)


class _TranscriptVtable(ctypes.Structure):     # zk_transcript_vtable
    _fields_ = [("common_point", _TR_IN), ("common_scalar", _TR_IN), ("write_point", _TR_IN), ("write_scalar", _TR_IN), ("squeeze_challenge", _TR_OUT)]


class ProofSession:
    """begin -> advice_phase(...) per phase (returns that phase's challenges) -> finish()."""

    def __init__(self, ctx: "Context", pk: ProvingKey, instance: Sequence[np.ndarray], seed: bytes, instance_slices: bool = False):
        """instance: (n, 4) column images (every usable row is absorbed), or -- instance_slices=True --
        halo2's instance slices as they are: (len_i, 4) arrays, exactly len_i values absorbed each."""
        self.ctx, self.pk = ctx, pk
        ins = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in instance]
        pi = (ctypes.c_void_p * max(len(ins), 1))(*[a.ctypes.data for a in ins])
        h = ctypes.c_void_p()
        assert len(seed) == 16
        if instance_slices:
            lens = (ctypes.c_uint32 * max(len(ins), 1))(*[a.shape[0] for a in ins])
            ctx._ck(lib().zk_proof_begin_instances(ctx.h, pk.h, pi, lens, ctypes.c_char_p(seed), ctypes.byref(h)))
        else:
            ctx._ck(lib().zk_proof_begin(ctx.h, pk.h, pi, ctypes.c_char_p(seed), ctypes.byref(h)))
        self.h = h

    def set_multiopen(self, kind: int):
        """0 = GWC (default), 1 = SHPLONK."""
        self.ctx._ck(lib().zk_proof_set_multiopen(self.ctx.h, self.h, ctypes.c_int(kind)))

    def set_vanishing_random(self, kind: int):
        """0 = n uniform coefficients (upstream PSE halo2), 1 = the constant 1 (the reference's own proofs; default)"""
        self.ctx._ck(lib().zk_proof_set_vanishing_random(self.ctx.h, self.h, ctypes.c_int(kind)))

    def set_transcript_kind(self, kind: int):
        """built-in transcript: TRANSCRIPT_BLAKE2B (default), TRANSCRIPT_POSEIDON or TRANSCRIPT_EVM"""
        self.ctx._ck(lib().zk_proof_set_transcript_kind(self.ctx.h, self.h, ctypes.c_int(kind)))

    def set_transcript(self, transcript):
        """Forward every transcript operation to `transcript`, an object with
        common_point(bytes64) / common_scalar(bytes32) / write_point(bytes64) / write_scalar(bytes32)
        (Montgomery limbs, as they cross the ABI) and squeeze_challenge() -> bytes32 (Montgomery Fr).
        The object then owns the proof bytes; finish() returns b''."""
        def wrap_in(fn, nbytes):
            def cb(_user, ptr):
                try:
                    fn(ctypes.string_at(ptr, nbytes))
                    return 0
                except Exception as e:       # never let an exception cross the C boundary
                    print(f"[zkmi355 transcript] callback failed: {e!r}", flush=True)
                    return 1
            return _TR_IN(cb)

        def squeeze(_user, out_ptr):
            try:
                ctypes.memmove(out_ptr, bytes(transcript.squeeze_challenge()), 32)
                return 0
            except Exception as e:
                print(f"[zkmi355 transcript] squeeze failed: {e!r}", flush=True)
                return 1
        vt = _TranscriptVtable(wrap_in(transcript.common_point, 64), wrap_in(transcript.common_scalar, 32),
                               wrap_in(transcript.write_point, 64), wrap_in(transcript.write_scalar, 32), _TR_OUT(squeeze))
        self._transcript = (transcript, vt)      # keep the thunks alive
        self.ctx._ck(lib().zk_proof_set_transcript(self.ctx.h, self.h, ctypes.byref(vt), None))

    def set_sharding(self, rank: int, world: int, allgather_cb):
        """This rank's share of a multi-GPU proof; allgather_cb is a sharding.ALLGATHER_FN instance."""
        self._gather_cb = allgather_cb          # keep the ctypes thunk alive
        self.ctx._ck(lib().zk_proof_set_sharding(self.ctx.h, self.h, ctypes.c_uint32(rank), ctypes.c_uint32(world), allgather_cb, None))

    def set_sharding_comm(self):
        """shard this session over the context's own RCCL communicator (Context.comm_init)"""
        self.ctx._ck(lib().zk_proof_set_sharding_comm(self.ctx.h, self.h))

    def set_device_gather(self, allgather_dev_cb):
        """After set_sharding: exchange the advice columns with a device all-gather (RCCL) instead of
        uploading every column on every rank; allgather_dev_cb is a sharding.ALLGATHER_FN over device pointers."""
        self._gather_dev_cb = allgather_dev_cb
        self.ctx._ck(lib().zk_proof_set_device_gather(self.ctx.h, self.h, allgather_dev_cb, None))

    def advice_phase(self, columns: dict) -> np.ndarray:
        """columns: {advice column index: (n, 4) u64 Montgomery array}; returns (num_challenges, 4) u64."""
        idx = sorted(columns)
        cols = [np.ascontiguousarray(columns[i], dtype=np.uint64) for i in idx]
        ci = (ctypes.c_uint32 * max(len(idx), 1))(*idx)
        pc = (ctypes.c_void_p * max(len(idx), 1))(*[c.ctypes.data for c in cols])
        cap = getattr(self, "_challenge_cap", None)
        if cap is None:
            cap = self._challenge_cap = max(1, self.pk.shape()["challenges"])      # sized from the key, never guessed
        out = np.zeros((cap, 4), dtype=np.uint64)
        cnt = ctypes.c_uint32(cap)
        self.ctx._ck(lib().zk_proof_advice_phase(self.ctx.h, self.h, ci, pc, ctypes.c_uint32(len(idx)), _host_ptr(out), ctypes.byref(cnt)))
        return out[:cnt.value].copy()

    def advice_phase_dev(self, columns: dict, in_place: bool = False) -> np.ndarray:
        """{advice column index: DeviceBuffer (n x 32 B, Montgomery)}: the phase's witness columns resident on the device.
        in_place: the session works in these buffers (it overwrites their blinding rows) until finish() / abort() returns.
        Sharded session with a device all-gather: a column this rank does not own (position j of the phase's columns in ascending
        index order, j % world != rank) may be None -- it arrives over the fabric from its owner."""
        idx = sorted(columns)
        ci = (ctypes.c_uint32 * max(len(idx), 1))(*idx)
        ptrs = (ctypes.c_void_p * max(len(idx), 1))(*[None if columns[i] is None else columns[i].ptr for i in idx])
        cap = getattr(self, "_challenge_cap", None)
        if cap is None:
            cap = self._challenge_cap = max(1, self.pk.shape()["challenges"])
        out = np.zeros((cap, 4), dtype=np.uint64)
        cnt = ctypes.c_uint32(cap)
        self.ctx._ck(lib().zk_proof_advice_phase_dev(self.ctx.h, self.h, ci, ptrs, ctypes.c_uint32(len(idx)), ctypes.c_uint32(1 if in_place else 0), _host_ptr(out), ctypes.byref(cnt)))
        return out[:cnt.value].copy()

    def mock_verify(self, gate_rows: Optional[Sequence[int]] = None, lookup_rows: Optional[Sequence[int]] = None, cap: int = 4096):
        """zk_proof_mock_verify: MockProver's row checks over the columns this session holds, with the challenges its transcript
        produced; after the last advice phase, before finish().  Returns (records, total) like Context.mock_verify."""
        gr = None if gate_rows is None else np.ascontiguousarray(gate_rows, dtype=np.uint32)
        lr = None if lookup_rows is None else np.ascontiguousarray(lookup_rows, dtype=np.uint32)
        out = np.zeros((max(cap, 1), 4), dtype=np.uint32)
        total = ctypes.c_size_t()
        self.ctx._ck(lib().zk_proof_mock_verify(self.ctx.h, self.h, None if gr is None else _host_ptr(gr), ctypes.c_size_t(0 if gr is None else len(gr)),
                                            None if lr is None else _host_ptr(lr), ctypes.c_size_t(0 if lr is None else len(lr)),
                                            _host_ptr(out), ctypes.c_size_t(cap), ctypes.byref(total)))
        return [tuple(int(v) for v in r) for r in out[:min(total.value, cap)]], total.value

    def finish(self) -> bytes:
        cap = 1 << 20
        out = ctypes.create_string_buffer(cap)
        n = ctypes.c_size_t()
        h, self.h = self.h, None
        self.ctx._ck(lib().zk_proof_finish(self.ctx.h, h, out, ctypes.c_size_t(cap), ctypes.byref(n)))
        return out.raw[:n.value]

    def abort(self):
        if self.h:
            lib().zk_proof_abort(self.ctx.h, self.h)
            self.h = None


class Context:
    """One zk_ctx: one GPU, one stream, one host thread."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        h = ctypes.c_void_p()
        rc = lib().zk_ctx_create(ctypes.c_int(device), ctypes.byref(h))
        if rc != 0:
            raise ZkError(f"zk_ctx_create failed with status {rc} (no gfx950 device? there is no CPU fallback)")
        self.h = h
        if stream is not None:
            self._ck(lib().zk_ctx_set_stream(self.h, ctypes.c_void_p(stream)))

    def _ck(self, rc: int):
        if rc != 0:
            raise ZkError(f"zkmi355 status {rc}: {lib().zk_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            lib().zk_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        """joins every stream of the library (main, copy, auxiliary, MSM side streams), not only the main one"""
        self._ck(lib().zk_ctx_sync(self.h))

    def streams_busy(self) -> int:
        n = ctypes.c_int()
        self._ck(lib().zk_ctx_streams_busy(self.h, ctypes.byref(n)))
        return n.value

    def debug_delay(self, role: int, usec: int):
        self._ck(lib().zk_ctx_debug_delay(self.h, ctypes.c_int(role), ctypes.c_uint32(usec)))

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu = ctypes.c_int()
        hbm = ctypes.c_size_t()
        self._ck(lib().zk_device_info(self.h, name, ctypes.c_size_t(256), ctypes.byref(cu), ctypes.byref(hbm)))
        return name.value.decode(), cu.value, hbm.value

    # ---- memory
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, a: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(a)
        return DeviceBuffer(self, max(a.nbytes, 1)).upload(a)

    def timer_start(self):
        self._ck(lib().zk_timer_start(self.h))

    def timer_stop_ms(self) -> float:
        ms = ctypes.c_float()
        self._ck(lib().zk_timer_stop_ms(self.h, ctypes.byref(ms)))
        return ms.value

    # ---- profiling
    def prof_enable(self, on=True):
        """True / 1: every kernel group; 2: only the roofline kernels' groups; False / 0: off"""
        self._ck(lib().zk_prof_enable(self.h, ctypes.c_int(int(on))))

    def prof_reset(self):
        self._ck(lib().zk_prof_reset(self.h))

    def prof_get(self, name: str):
        ms, cnt = ctypes.c_double(), ctypes.c_uint64()
        self._ck(lib().zk_prof_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def prof_get_bytes(self, name: str) -> int:
        b = ctypes.c_uint64()
        self._ck(lib().zk_prof_get_bytes(self.h, name.encode(), ctypes.byref(b)))
        return b.value

    def prof_names(self):
        buf = ctypes.create_string_buffer(4096)
        self._ck(lib().zk_prof_names(self.h, buf, ctypes.c_size_t(4096)))
        return [x for x in buf.value.decode().split(";") if x]

    # ---- field vectors
    def field_vec_op(self, field: int, op: int, a: DeviceBuffer, b: DeviceBuffer, out: DeviceBuffer, n: int):
        self._ck(lib().zk_field_vec_op(self.h, field, op, ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(out.ptr), ctypes.c_size_t(n)))

    def fr_scale(self, a: DeviceBuffer, s: np.ndarray, n: int):
        self._ck(lib().zk_fr_scale(self.h, ctypes.c_void_p(a.ptr), _host_ptr(np.ascontiguousarray(s)), ctypes.c_size_t(n)))

    def fr_batch_invert(self, a: DeviceBuffer, n: int):
        self._ck(lib().zk_fr_batch_invert(self.h, ctypes.c_void_p(a.ptr), ctypes.c_size_t(n)))

    def fr_prefix_product(self, a: DeviceBuffer, z: DeviceBuffer, n: int):
        self._ck(lib().zk_fr_prefix_product(self.h, ctypes.c_void_p(a.ptr), ctypes.c_void_p(z.ptr), ctypes.c_size_t(n)))

    def fr_prefix_sum(self, a: DeviceBuffer, z: DeviceBuffer, n: int):
        self._ck(lib().zk_fr_prefix_sum(self.h, ctypes.c_void_p(a.ptr), ctypes.c_void_p(z.ptr), ctypes.c_size_t(n)))

    # ---- NTT (halo2 best_fft / EvaluationDomain)
    def ntt(self, data: DeviceBuffer, log_n: int, inverse: bool = False):
        self._ck(lib().zk_ntt(self.h, ctypes.c_void_p(data.ptr), ctypes.c_uint32(log_n), ctypes.c_int(1 if inverse else 0)))

    def ntt_batch(self, datas: Sequence[DeviceBuffer], log_n: int, inverse: bool = False):
        """zk_ntt over several columns of the same size, several columns per launch"""
        ptrs = (ctypes.c_void_p * max(len(datas), 1))(*[ctypes.c_void_p(d.ptr) for d in datas])
        self._ck(lib().zk_ntt_batch(self.h, ptrs, ctypes.c_size_t(len(datas)), ctypes.c_uint32(log_n), ctypes.c_int(1 if inverse else 0)))

    def coeff_to_coset_batch(self, coeffs: Sequence[DeviceBuffer], k: int, g_mont: np.ndarray, outs: Sequence[DeviceBuffer]):
        cp = (ctypes.c_void_p * max(len(coeffs), 1))(*[ctypes.c_void_p(d.ptr) for d in coeffs])
        op = (ctypes.c_void_p * max(len(outs), 1))(*[ctypes.c_void_p(d.ptr) for d in outs])
        self._ck(lib().zk_coeff_to_coset_batch(self.h, cp, ctypes.c_uint32(k), _host_ptr(np.ascontiguousarray(g_mont)), op, ctypes.c_size_t(len(coeffs))))

    def ntt_sharded(self, local: DeviceBuffer, log_n: int, rank: int, world: int, alltoall_cb, inverse: bool = False):
        """This rank's part of ONE 2^log_n transform spread over `world` GPUs (zk_ntt_sharded):
        in: x[rank + world * i]; out: [j1][c] = X[(rank * m / world + c) + m * j1], m = n / world.
        alltoall_cb is a sharding.ALLTOALL_FN over device pointers."""
        self._ck(lib().zk_ntt_sharded(self.h, ctypes.c_void_p(local.ptr), ctypes.c_uint32(log_n), ctypes.c_int(1 if inverse else 0),
                                      ctypes.c_uint32(rank), ctypes.c_uint32(world), alltoall_cb, None))

    def ntt_omega(self, data: DeviceBuffer, log_n: int, omega_mont: np.ndarray):
        self._ck(lib().zk_ntt_omega(self.h, ctypes.c_void_p(data.ptr), ctypes.c_uint32(log_n), _host_ptr(np.ascontiguousarray(omega_mont))))

    def coeff_to_extended(self, coeffs: DeviceBuffer, k: int, ext_k: int, out: DeviceBuffer):
        self._ck(lib().zk_coeff_to_extended(self.h, ctypes.c_void_p(coeffs.ptr), ctypes.c_uint32(k), ctypes.c_uint32(ext_k), ctypes.c_void_p(out.ptr)))

    def coeff_to_coset(self, coeffs: DeviceBuffer, k: int, g_mont: np.ndarray, out: DeviceBuffer):
        """out[i] = f(g * omega^i): one coset of the extended domain (coeff_to_extended_part)."""
        self._ck(lib().zk_coeff_to_coset(self.h, ctypes.c_void_p(coeffs.ptr), ctypes.c_uint32(k), _host_ptr(np.ascontiguousarray(g_mont)), ctypes.c_void_p(out.ptr)))

    def fr_scatter_scaled(self, src: DeviceBuffer, n: int, scale_mont: np.ndarray, dst: DeviceBuffer, stride: int, offset: int):
        self._ck(lib().zk_fr_scatter_scaled(self.h, ctypes.c_void_p(src.ptr), ctypes.c_size_t(n), _host_ptr(np.ascontiguousarray(scale_mont)),
                                            ctypes.c_void_p(dst.ptr), ctypes.c_size_t(stride), ctypes.c_size_t(offset)))

    def extended_to_coeff(self, ext: DeviceBuffer, ext_k: int):
        self._ck(lib().zk_extended_to_coeff(self.h, ctypes.c_void_p(ext.ptr), ctypes.c_uint32(ext_k)))

    # ---- polynomial helpers
    def poly_eval(self, coeffs: DeviceBuffer, n: int, x_mont: np.ndarray) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        self._ck(lib().zk_poly_eval(self.h, ctypes.c_void_p(coeffs.ptr), ctypes.c_size_t(n), _host_ptr(np.ascontiguousarray(x_mont)), _host_ptr(out)))
        return out

    def poly_eval_batch(self, polys, n: int, x_mont: np.ndarray) -> np.ndarray:
        """eval_polynomial of several device polynomials (n coefficients each) at one point -> (count, 4) u64."""
        count = len(polys)
        out = np.empty((count, 4), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * max(count, 1))(*[ctypes.c_void_p(p.ptr) for p in polys])
        self._ck(lib().zk_poly_eval_batch(self.h, ptrs, ctypes.c_size_t(count), ctypes.c_size_t(n), _host_ptr(np.ascontiguousarray(x_mont)), _host_ptr(out)))
        return out

    def poly_eval_pairs(self, polys, point_index: Sequence[int], points_mont: np.ndarray, n: int) -> np.ndarray:
        """zk_poly_eval_pairs: polys[j] (device, n coefficients) at points_mont[point_index[j]] -> (count, 4) u64, one pass"""
        count = len(polys)
        out = np.empty((max(count, 1), 4), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * max(count, 1))(*[ctypes.c_void_p(p.ptr) for p in polys])
        idx = np.ascontiguousarray(point_index, dtype=np.uint32)
        pts = np.ascontiguousarray(points_mont, dtype=np.uint64).reshape(-1, 4)
        self._ck(lib().zk_poly_eval_pairs(self.h, ptrs, _host_ptr(idx), ctypes.c_size_t(count), _host_ptr(pts), ctypes.c_size_t(len(pts)), ctypes.c_size_t(n), _host_ptr(out)))
        return out[:count]

    def fr_random(self, key32: bytes, stream_id: int, first_block: int, out: DeviceBuffer, n: int):
        """n uniform Fr from ChaCha20 blocks (counter mode) + from_uniform_bytes."""
        assert len(key32) == 32
        self._ck(lib().zk_fr_random(self.h, ctypes.c_char_p(key32), ctypes.c_uint64(stream_id), ctypes.c_uint64(first_block), ctypes.c_void_p(out.ptr), ctypes.c_size_t(n)))

    def host_alloc(self, shape, dtype=np.uint64) -> np.ndarray:
        """numpy array over page-locked host memory (zk_host_alloc); freed with host_free(array)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self._ck(lib().zk_host_alloc(self.h, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._ck(lib().zk_host_free(self.h, ctypes.c_void_p(p)))

    def host_register(self, arr: np.ndarray):
        """Page-lock an array the caller owns (zk_host_register); pair with host_unregister(arr)."""
        assert arr.flags["C_CONTIGUOUS"]
        self._ck(lib().zk_host_register(self.h, ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes)))

    def host_unregister(self, arr: np.ndarray):
        self._ck(lib().zk_host_unregister(self.h, ctypes.c_void_p(arr.ctypes.data)))

    # ---- in-library RCCL collectives (csrc/comm.hip)
    @staticmethod
    def comm_unique_id() -> bytes:
        out = ctypes.create_string_buffer(128)
        rc = lib().zk_comm_unique_id(out)
        if rc != 0:
            raise ZkError(f"zk_comm_unique_id failed with status {rc} (librccl missing?)")
        return out.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._ck(lib().zk_comm_init(self.h, ctypes.c_char_p(unique_id), ctypes.c_uint32(rank), ctypes.c_uint32(world)))

    def comm_destroy(self):
        self._ck(lib().zk_comm_destroy(self.h))

    def comm_allgather(self, send: DeviceBuffer, nbytes: int, recv: DeviceBuffer):
        self._ck(lib().zk_comm_allgather(self.h, ctypes.c_void_p(send.ptr), ctypes.c_size_t(nbytes), ctypes.c_void_p(recv.ptr)))

    def comm_alltoall(self, send: DeviceBuffer, bytes_per_peer: int, recv: DeviceBuffer):
        self._ck(lib().zk_comm_alltoall(self.h, ctypes.c_void_p(send.ptr), ctypes.c_size_t(bytes_per_peer), ctypes.c_void_p(recv.ptr)))

    def msm_plan(self, srs, n: int) -> dict:
        c, w = ctypes.c_int(), ctypes.c_int()
        self._ck(lib().zk_msm_plan(srs.h, ctypes.c_size_t(n), ctypes.byref(c), ctypes.byref(w)))
        return {"c": c.value, "windows": w.value}

    def lookup_multiplicities(self, inputs: DeviceBuffer, table: DeviceBuffer, usable_rows: int, m: DeviceBuffer, n: int) -> Optional[int]:
        """logUp m(X) on the device; returns the lowest input row missing from the table, or None."""
        bad = ctypes.c_uint64()
        self._ck(lib().zk_lookup_multiplicities(self.h, ctypes.c_void_p(inputs.ptr), ctypes.c_void_p(table.ptr), ctypes.c_size_t(usable_rows),
                                                ctypes.c_void_p(m.ptr), ctypes.c_size_t(n), ctypes.byref(bad)))
        return None if bad.value == 0xFFFFFFFFFFFFFFFF else bad.value

    def commit_batch_h2d(self, srs: "Srs", basis: int, host_cols, dev_cols, n: int) -> np.ndarray:
        """commit `host_cols` (numpy (n,4) u64 each) while uploading them into dev_cols (overlapped)."""
        count = len(host_cols)
        hc = [np.ascontiguousarray(c, dtype=np.uint64) for c in host_cols]
        hp = (ctypes.c_void_p * max(count, 1))(*[ctypes.c_void_p(c.ctypes.data) for c in hc])
        dp = (ctypes.c_void_p * max(count, 1))(*[ctypes.c_void_p(d.ptr) for d in dev_cols])
        out = np.empty((count, 8), dtype=np.uint64)
        self._ck(lib().zk_commit_batch_h2d(self.h, srs.h, ctypes.c_int(basis), hp, dp, ctypes.c_size_t(count), ctypes.c_size_t(n), _host_ptr(out)))
        return out

    def kate_division(self, coeffs: DeviceBuffer, n: int, z_mont: np.ndarray, q: DeviceBuffer):
        self._ck(lib().zk_kate_division(self.h, ctypes.c_void_p(coeffs.ptr), ctypes.c_size_t(n), _host_ptr(np.ascontiguousarray(z_mont)), ctypes.c_void_p(q.ptr)))

    # ---- expression / quotient evaluation (halo2 evaluate_h)
    def quotient_eval(self, program: np.ndarray, col_ptrs, consts: np.ndarray, k: int, ext_k: int, out: DeviceBuffer, divide_by_vanishing: bool = False):
        prog = np.ascontiguousarray(program, dtype=np.uint32).reshape(-1, 3)
        ptrs = (ctypes.c_void_p * max(len(col_ptrs), 1))(*[ctypes.c_void_p(p) for p in col_ptrs])
        consts = np.ascontiguousarray(consts, dtype=np.uint64).reshape(-1, 4)
        self._ck(lib().zk_quotient_eval(self.h, prog.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(prog.shape[0]), ptrs, ctypes.c_uint32(len(col_ptrs)),
                                        _host_ptr(consts) if consts.size else None, ctypes.c_uint32(consts.shape[0]), ctypes.c_uint32(k), ctypes.c_uint32(ext_k),
                                        ctypes.c_int(1 if divide_by_vanishing else 0), ctypes.c_void_p(out.ptr)))

    def fr_powers(self, base_mont: np.ndarray, mul_mont: np.ndarray, out: DeviceBuffer, n: int):
        self._ck(lib().zk_fr_powers(self.h, _host_ptr(np.ascontiguousarray(base_mont)), _host_ptr(np.ascontiguousarray(mul_mont)), ctypes.c_void_p(out.ptr), ctypes.c_size_t(n)))

    # ---- SRS / MSM (halo2 ParamsKZG / best_multiexp)
    def srs_create(self, k: int, g: np.ndarray, g_lagrange: Optional[np.ndarray] = None) -> Srs:
        h = ctypes.c_void_p()
        gl = _host_ptr(np.ascontiguousarray(g_lagrange)) if g_lagrange is not None else None
        self._ck(lib().zk_srs_create(self.h, ctypes.c_uint32(k), _host_ptr(np.ascontiguousarray(g)), gl, ctypes.byref(h)))
        return Srs(self, h)

    # ---- SRS files (ParamsKZG::read_custom / write_custom): format 0 Processed, 1 RawBytes, 2 RawBytesUnchecked
    def params_read(self, data: bytes, fmt: int = 2):
        """-> (Srs, g2 bytes, s_g2 bytes); refuses a length that does not match the header's k."""
        h = ctypes.c_void_p()
        g2len = 64 if fmt == 0 else 128
        g2, sg2 = ctypes.create_string_buffer(g2len), ctypes.create_string_buffer(g2len)
        buf = np.frombuffer(data, dtype=np.uint8)
        self._ck(lib().zk_params_read(self.h, _host_ptr(buf), ctypes.c_size_t(buf.size), ctypes.c_int(fmt), ctypes.byref(h), g2, sg2))
        return Srs(self, h), g2.raw, sg2.raw

    def params_write(self, srs: Srs, g2: bytes, s_g2: bytes, fmt: int = 2) -> bytes:
        need = ctypes.c_size_t()
        self._ck(lib().zk_params_write(self.h, srs.h, g2, s_g2, ctypes.c_int(fmt), None, ctypes.c_size_t(0), ctypes.byref(need)))
        out = np.empty(need.value, dtype=np.uint8)
        self._ck(lib().zk_params_write(self.h, srs.h, g2, s_g2, ctypes.c_int(fmt), _host_ptr(out), ctypes.c_size_t(out.size), ctypes.byref(need)))
        return out.tobytes()

    def srs_setup_with_s(self, k: int, s_mont: np.ndarray) -> Srs:
        h = ctypes.c_void_p()
        self._ck(lib().zk_srs_setup_with_s(self.h, ctypes.c_uint32(k), _host_ptr(np.ascontiguousarray(s_mont)), ctypes.byref(h)))
        return Srs(self, h)

    def msm(self, scalars_ptr: int, bases_ptr: int, n: int) -> np.ndarray:
        out = np.empty(8, dtype=np.uint64)
        self._ck(lib().zk_msm_g1(self.h, ctypes.c_void_p(scalars_ptr), ctypes.c_void_p(bases_ptr), ctypes.c_size_t(n), _host_ptr(out)))
        return out

    def commit(self, srs: Srs, scalars: DeviceBuffer, n: int, lagrange: bool = False) -> np.ndarray:
        out = np.empty(8, dtype=np.uint64)
        self._ck(lib().zk_commit(self.h, srs.h, ctypes.c_int(1 if lagrange else 0), ctypes.c_void_p(scalars.ptr), ctypes.c_size_t(n), _host_ptr(out)))
        return out

    def commit_batch(self, srs: Srs, scalar_ptrs: Sequence[int], n: int, lagrange: bool = False, narrow: Optional[Sequence[int]] = None) -> np.ndarray:
        """`len(scalar_ptrs)` commitments over one basis, pipelined on the device; returns (count, 8) u64.
        narrow: optional per-column hint (0 dense, 1 small integers: per-window MSM path, 2 long runs of equal
        values: sliced sort), speed only."""
        count = len(scalar_ptrs)
        out = np.empty((max(count, 1), 8), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * max(count, 1))(*[ctypes.c_void_p(p) for p in scalar_ptrs])
        if narrow is None:
            self._ck(lib().zk_commit_batch(self.h, srs.h, ctypes.c_int(1 if lagrange else 0), ptrs, ctypes.c_size_t(count), ctypes.c_size_t(n), _host_ptr(out)))
        else:
            assert len(narrow) == count
            flags = (ctypes.c_uint8 * max(count, 1))(*[int(f) for f in narrow])
            self._ck(lib().zk_commit_batch_hint(self.h, srs.h, ctypes.c_int(1 if lagrange else 0), ptrs, ctypes.c_size_t(count), ctypes.c_size_t(n), flags, _host_ptr(out)))
        return out[:count]

    def best_multiexp(self, scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
        """halo2 ``best_multiexp(coeffs, bases)`` over host slices."""
        scalars = np.ascontiguousarray(scalars)
        bases = np.ascontiguousarray(bases)
        n = scalars.shape[0] if scalars.size else 0
        out = np.empty(8, dtype=np.uint64)
        self._ck(lib().zk_msm_g1_host(self.h, _host_ptr(scalars), _host_ptr(bases), ctypes.c_size_t(n), _host_ptr(out)))
        return out

    def best_fft(self, a: np.ndarray, log_n: int, inverse: bool = False) -> np.ndarray:
        """halo2 ``best_fft`` over a host slice (copy in, transform, copy out)."""
        buf = self.to_device(a)
        self.ntt(buf, log_n, inverse)
        out = buf.download(a.shape)
        buf.free()
        return out

    # ---- full proofs (halo2 keygen_pk / create_proof, GWC)
    def pk_create(self, srs: Srs, blob: bytes) -> "ProvingKey":
        h = ctypes.c_void_p()
        if isinstance(blob, np.ndarray):      # large keys: a uint8 array is passed by pointer, no copy
            assert blob.dtype == np.uint8 and blob.flags["C_CONTIGUOUS"]
            self._ck(lib().zk_pk_create(self.h, srs.h, _host_ptr(blob), ctypes.c_size_t(blob.nbytes), ctypes.byref(h)))
            return ProvingKey(self, h)
        view = np.frombuffer(blob, dtype=np.uint8)      # read-only view of the bytes object: no copy of a multi-GiB key
        self._ck(lib().zk_pk_create(self.h, srs.h, _host_ptr(view), ctypes.c_size_t(view.size), ctypes.byref(h)))
        return ProvingKey(self, h)

    def create_proof(self, pk: "ProvingKey", advice: Sequence[np.ndarray], instance: Sequence[np.ndarray], seed: bytes = bytes(16)) -> bytes:
        adv = [np.ascontiguousarray(a, dtype=np.uint64) for a in advice]
        ins = [np.ascontiguousarray(a, dtype=np.uint64) for a in instance]
        pa = (ctypes.c_void_p * max(len(adv), 1))(*[a.ctypes.data for a in adv])
        pi = (ctypes.c_void_p * max(len(ins), 1))(*[a.ctypes.data for a in ins])
        cap = 1 << 20
        out = ctypes.create_string_buffer(cap)
        n = ctypes.c_size_t()
        assert len(seed) == 16
        self._ck(lib().zk_create_proof(self.h, pk.h, pa, pi, ctypes.c_char_p(seed), out, ctypes.c_size_t(cap), ctypes.byref(n)))
        return out.raw[:n.value]

    def mock_verify(self, pk: "ProvingKey", advice: Sequence[np.ndarray], instance: Sequence[np.ndarray], challenges: Optional[np.ndarray] = None,
                    gate_rows: Optional[Sequence[int]] = None, lookup_rows: Optional[Sequence[int]] = None, cap: int = 4096):
        """zk_mock_verify (halo2 dev::MockProver::verify_par / verify_at_rows_par): (failures, total) with failures the first
        `cap` records (kind, index, sub, row), sorted; kinds 1 gate, 2 lookup, 3 permutation.  challenges: (c, 4) u64
        Montgomery or None for MockProver's own."""
        adv = [np.ascontiguousarray(a, dtype=np.uint64) for a in advice]
        ins = [np.ascontiguousarray(a, dtype=np.uint64) for a in instance]
        pa = (ctypes.c_void_p * max(len(adv), 1))(*[a.ctypes.data for a in adv])
        pi = (ctypes.c_void_p * max(len(ins), 1))(*[a.ctypes.data for a in ins])
        ch = None if challenges is None else np.ascontiguousarray(challenges, dtype=np.uint64)
        gr = None if gate_rows is None else np.ascontiguousarray(gate_rows, dtype=np.uint32)
        lr = None if lookup_rows is None else np.ascontiguousarray(lookup_rows, dtype=np.uint32)
        out = np.zeros((max(cap, 1), 4), dtype=np.uint32)
        total = ctypes.c_size_t()
        self._ck(lib().zk_mock_verify(self.h, pk.h, pa, pi, None if ch is None else _host_ptr(ch),
                                      None if gr is None else _host_ptr(gr), ctypes.c_size_t(0 if gr is None else len(gr)),
                                      None if lr is None else _host_ptr(lr), ctypes.c_size_t(0 if lr is None else len(lr)),
                                      _host_ptr(out), ctypes.c_size_t(cap), ctypes.byref(total)))
        return [tuple(int(v) for v in r) for r in out[:min(total.value, cap)]], total.value

    def proof_session(self, pk: "ProvingKey", instance: Sequence[np.ndarray], seed: bytes = bytes(16), instance_slices: bool = False) -> "ProofSession":
        return ProofSession(self, pk, instance, seed, instance_slices)

    # ---- G1 element-wise
    def g1_affine_add(self, a: DeviceBuffer, b: DeviceBuffer, out: DeviceBuffer, n: int):
        self._ck(lib().zk_g1_affine_add_vec(self.h, ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(out.ptr), ctypes.c_size_t(n)))

    def g1_mul(self, bases: DeviceBuffer, scalars: DeviceBuffer, out: DeviceBuffer, n: int):
        self._ck(lib().zk_g1_mul_vec(self.h, ctypes.c_void_p(bases.ptr), ctypes.c_void_p(scalars.ptr), ctypes.c_void_p(out.ptr), ctypes.c_size_t(n)))


def mock_challenges(count: int) -> np.ndarray:
    """zk_host_mock_challenges: the challenges halo2's MockProver hands a circuit, (count, 4) u64 Montgomery Fr (host only)"""
    out = np.zeros((max(count, 1), 4), dtype=np.uint64)
    rc = lib().zk_host_mock_challenges(ctypes.c_uint32(count), _host_ptr(out))
    if rc != 0:
        raise ZkError(f"zk_host_mock_challenges failed with status {rc}")
    return out[:count]


def version() -> str:
    return lib().zk_version().decode()


def g2_setup(s_mont: np.ndarray):
    """unsafe_setup_with_s, G2 half: (generator, s * generator) as 128-byte RawBytes each (host only)."""
    g2, sg2 = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    rc = lib().zk_g2_setup(_host_ptr(np.ascontiguousarray(s_mont)), g2, sg2)
    if rc != 0:
        raise ZkError(f"zk_g2_setup failed: {rc}")
    return g2.raw, sg2.raw


def params_file_len(k: int, fmt: int = 2) -> int:
    return int(lib().zk_params_file_len(ctypes.c_uint32(k), ctypes.c_int(fmt)))
