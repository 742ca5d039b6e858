"""Global mode: ONE suffix array of a text sharded over P ranks (include/dc3hip.h, "GLOBAL mode").

    GlobalRank      one rank of a group (RCCL: one process per GPU; the 128-byte id travels through the caller's
                    own channel, e.g. torch.distributed.broadcast_object_list)
    LoopbackGroup   P ranks on ONE device inside this process — the parity-test vehicle for P in {2,4,8}

The reference has no global mode (sacapart keeps P independent arrays, crates/sacapart/src/lib.rs:5-25); the result is
defined by the single-device one: the shards in rank order concatenate to SA[0..n) of the whole text."""
import ctypes

import numpy as np

from ._lib import Dc3HipError, GStats, Stats, lib
from .api import _as_u8, _check


def torch_host_callbacks(dist, rank, nranks):
    """The two collectives of dc3hip_host_transport on raw host addresses, over torch.distributed (CPU tensors):
    all_to_all_v by batched isend/irecv, all_gather_v by all_gather of blocks padded to the largest one.
    Return 0 / 1 (a Python exception must not unwind through the C frames of the caller)."""
    import torch

    def view(ptr, nbytes):
        return torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

    def a2a(user, send, soff, sbytes, recv, roff, rbytes):
        try:
            ops, keep = [], []
            for r in range(nranks):
                if r == rank:
                    if rbytes[r]:
                        ctypes.memmove(recv + roff[r], send + soff[r], rbytes[r])
                    continue
                if sbytes[r]:
                    t = view(send + soff[r], sbytes[r]); keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, r))
                if rbytes[r]:
                    t = view(recv + roff[r], rbytes[r]); keep.append(t)
                    ops.append(dist.P2POp(dist.irecv, t, r))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return 0
        except Exception as e:
            print("host transport all_to_all_v:", repr(e), flush=True)
            return 1

    def ag(user, send, sbytes, recv, roff, rbytes):
        try:
            mx = max(max(int(rbytes[r]) for r in range(nranks)), 1)
            mine = torch.zeros(mx, dtype=torch.uint8)
            if sbytes:
                mine[:sbytes] = view(send, sbytes)
            outs = [torch.zeros(mx, dtype=torch.uint8) for _ in range(nranks)]
            dist.all_gather(outs, mine)
            for r in range(nranks):
                if rbytes[r]:
                    view(recv + roff[r], rbytes[r]).copy_(outs[r][:rbytes[r]])
            return 0
        except Exception as e:
            print("host transport all_gather_v:", repr(e), flush=True)
            return 1

    return a2a, ag


def global_plan(total_n, nranks):
    """Per-rank HBM need of a global-mode build (dc3hip_global_plan: the library's own sizing rules, no device touched)."""
    from ._lib import GPlan
    p = GPlan()
    _check(lib().dc3hip_global_plan(total_n, nranks, ctypes.byref(p)))
    return p.as_dict()


def block_of(total_n, nranks, rank):
    """(offset, length) of the text block rank `rank` owns: sacapart-style blocks of len/P + 1 bytes
    (crates/sacapart/src/lib.rs:43-46), the same arithmetic as dc3hip_global_block."""
    S = total_n // nranks + 1
    off = min(total_n, rank * S)
    return off, min(S, total_n - off)


class GlobalRank:
    def __init__(self, handle):
        self._h = ctypes.c_void_p(handle)
        self.total_n = 0

    @classmethod
    def rccl(cls, unique_id: bytes, rank: int, nranks: int, device: int, max_total_n: int):
        assert len(unique_id) == 128
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().dc3hip_global_rccl_create(ctypes.byref(h), buf, rank, nranks, device, max_total_n))
        return cls(h.value)

    @classmethod
    def torch_host(cls, dist, rank: int, nranks: int, device: int, max_total_n: int):
        """Host-staged transport over a torch.distributed CPU process group (gloo): multi-process runs on a single GPU
        and nodes without peer access.  `dist` = the initialised torch.distributed module."""
        from ._lib import A2A_FN, AG_FN, HostTransport
        a2a, ag = torch_host_callbacks(dist, rank, nranks)
        tr = HostTransport(None, A2A_FN(a2a), AG_FN(ag))
        h = ctypes.c_void_p()
        _check(lib().dc3hip_global_host_create(ctypes.byref(h), ctypes.byref(tr), rank, nranks, device, max_total_n))
        self = cls(h.value)
        self._transport_keepalive = tr          # the library keeps the function pointers
        return self

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        _check(lib().dc3hip_rccl_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def rccl_library():
        """(path of the RCCL the library bound to, True if the host program had it mapped before the library asked)."""
        buf = ctypes.create_string_buffer(4096)
        pre = ctypes.c_int32(0)
        _check(lib().dc3hip_rccl_library_path(buf, 4096, ctypes.byref(pre)))
        return buf.value.decode(), bool(pre.value)

    def close(self):
        if self._h:
            lib().dc3hip_global_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def block(self, total_n):
        off, ln = ctypes.c_int64(), ctypes.c_int64()
        _check(lib().dc3hip_global_block(self._h, total_n, ctypes.byref(off), ctypes.byref(ln)))
        return off.value, ln.value

    def set_text_block(self, block, total_n):
        b = _as_u8(block)
        off, ln = self.block(total_n)
        assert len(b) == ln, f"this rank's block of a {total_n}-byte text is {ln} bytes, got {len(b)}"
        _check(lib().dc3hip_global_set_text_block(self._h, b.ctypes.data if ln else None, total_n))
        self.total_n = total_n

    def generate(self, total_n, seed, kind=0):
        _check(lib().dc3hip_global_generate(self._h, total_n, seed, kind))
        self.total_n = total_n

    def build(self):
        rc = lib().dc3hip_global_build(self._h)
        if rc != 0:
            raise Dc3HipError(rc, lib().dc3hip_global_last_error(self._h).decode())

    def shard(self):
        first, cnt = ctypes.c_int64(), ctypes.c_int64()
        _check(lib().dc3hip_global_shard(self._h, ctypes.byref(first), ctypes.byref(cnt)))
        return first.value, cnt.value

    def shard_sa(self, dtype=np.int64):
        first, cnt = self.shard()
        if dtype == np.int64:
            out = np.zeros(cnt, dtype=np.int64)
            _check(lib().dc3hip_global_get_shard_i64(self._h, out.ctypes.data if cnt else None))
        else:
            out = np.zeros(cnt, dtype=np.uint32)
            _check(lib().dc3hip_global_get_shard_u32(self._h, out.ctypes.data if cnt else None))
        return first, out

    def shard_checksum(self):
        v = ctypes.c_uint64()
        _check(lib().dc3hip_global_shard_checksum(self._h, ctypes.byref(v)))
        return int(v.value)

    def sufcheck(self):
        """Wide contexts (64-bit positions): the collective verifier — every rank must call it.  0 = the concatenated
        shards are the suffix array, -2 / -3 = range / order violation."""
        rc = lib().dc3hip_global_sufcheck(self._h)
        if rc <= -11:
            raise Dc3HipError(rc + 10, lib().dc3hip_global_last_error(self._h).decode())
        return rc

    def stats(self):
        g, s = GStats(), Stats()
        _check(lib().dc3hip_global_stats(self._h, ctypes.byref(g), ctypes.byref(s)))
        d = g.as_dict()
        d["ctx"] = s.as_dict()
        return d

    def transport(self):
        return lib().dc3hip_global_transport(self._h).decode()

    def selftest(self):
        """Transport self-test, a COLLECTIVE (every rank calls it): ragged all-to-all / all-gather of known bytes, every
        byte checked.  Returns the number of ranks the transport itself reports (RCCL: ncclCommCount); raises on a
        wrong byte."""
        cnt = ctypes.c_int32(0)
        rc = lib().dc3hip_global_selftest(self._h, ctypes.byref(cnt))
        if rc != 0:
            raise Dc3HipError(rc, lib().dc3hip_global_last_error(self._h).decode())
        return int(cnt.value)


DEVICE_SPREAD = -2      # DC3HIP_DEVICE_SPREAD: rank r on device r % (visible devices)


class LoopbackGroup:
    """P ranks in this process, run on P host threads inside the library (dc3hip_global_loopback_build): all on one
    device (tests), or with device=DEVICE_SPREAD one per visible GPU — a single process using the whole node."""

    def __init__(self, nranks, max_total_n, device=-1):
        self.P = nranks
        self._arr = (ctypes.c_void_p * nranks)()
        _check(lib().dc3hip_global_loopback_create(self._arr, nranks, device, max_total_n))
        self.ranks = [GlobalRank(self._arr[r]) for r in range(nranks)]

    def close(self):
        for r in self.ranks:
            r.close()
        self.ranks = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_text(self, text):
        t = _as_u8(text)
        for r in self.ranks:
            off, ln = r.block(len(t))
            r.set_text_block(t[off:off + ln], len(t))

    def generate(self, total_n, seed, kind=0):
        for r in self.ranks:
            r.generate(total_n, seed, kind)

    def build(self):
        rc = lib().dc3hip_global_loopback_build(self._arr, self.P)
        if rc != 0:
            from .api import last_error
            raise Dc3HipError(rc, last_error())

    def sa(self):
        """The shards in rank order, concatenated (int64); also checks that they tile [0, n)."""
        parts, nxt = [], 0
        for r in self.ranks:
            first, s = r.shard_sa(np.int64)
            assert first == nxt, f"shard of rank starts at {first}, expected {nxt}"
            nxt += len(s)
            parts.append(s)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)

    def checksum(self):
        return sum(r.shard_checksum() for r in self.ranks) & (2**64 - 1)

    def sufcheck(self):
        """Wide groups: dc3hip_global_sufcheck of all ranks (a collective: one host thread per rank)."""
        import threading
        out = [None] * self.P

        def run(i):
            try:
                out[i] = self.ranks[i].sufcheck()
            except Exception as e:          # noqa: BLE001 - reported below
                out[i] = e
        th = [threading.Thread(target=run, args=(i,)) for i in range(self.P)]
        for t in th: t.start()
        for t in th: t.join()
        for v in out:
            if isinstance(v, Exception):
                raise v
        assert len(set(out)) == 1, out
        return out[0]

    def stats(self):
        return [r.stats() for r in self.ranks]

    def _collective(self, fn):
        import threading
        out = [None] * self.P

        def run(i):
            try:
                out[i] = fn(self.ranks[i])
            except Exception as e:          # noqa: BLE001 - reported below
                out[i] = e
        th = [threading.Thread(target=run, args=(i,)) for i in range(self.P)]
        for t in th: t.start()
        for t in th: t.join()
        for v in out:
            if isinstance(v, Exception):
                raise v
        return out

    def selftest(self):
        """dc3hip_global_selftest of all ranks (a collective: one host thread per rank)."""
        return self._collective(lambda r: r.selftest())
