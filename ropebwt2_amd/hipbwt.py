"""ctypes mirror of include/rb2_hip.h -- the C ABI of the gfx950 insertion engine.

``HipBwt`` follows the reference's operator interface for this path: ``insert_multi(buf)`` takes
exactly the buffer main.c hands to ``mr_insert_multi`` (mrope.h:46-54: concatenated, reversed,
0-terminated nt6 strings) and mutates the six ropes; ``counts()`` is ``rope_t.c`` (rope.h:19).
"""
import ctypes as C
import os

import numpy as np

from .build import lib_path

K_NAMES = ["k_sym", "k_tscan", "k_prep", "k_part", "k_merge", "k_meta", "k_advance", "k_init", "k_relayout", "k_split"]
_lib = None


def load_hip_lib():
    """Load librb2hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RB2_HIP_LIB") or lib_path("librb2hip.so")      # (RB2_HIP_LIB: A/B runs of two builds on one box)
    if not os.path.exists(path):
        raise RuntimeError("%s missing: run `python -m ropebwt2_amd.build` (hipcc, gfx950) first" % path)
    L = C.CDLL(path)
    vp, i64, i32, u64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64
    sig = {
        "rb2_hip_device_count": (i32, []),
        "rb2_hip_set_fatal_handler": (None, [vp, vp]),
        "rb2_hip_create": (vp, [i32, i32]),
        "rb2_hip_destroy": (None, [vp]),
        "rb2_hip_sorting_order": (i32, [vp]),
        "rb2_hip_reset": (None, [vp]),
        "rb2_hip_insert_multi": (None, [vp, i64, vp]),
        "rb2_hip_insert_multi_dev": (None, [vp, i64, vp]),
        "rb2_hip_prefetch": (None, [vp, vp, i64, i64]),
        "rb2_hip_set_lazy": (None, [vp, C.c_int]),
        "rb2_hip_wait": (None, [vp]),
        "rb2_hip_last_batch_counts": (C.c_int, [vp, vp]),
        "rb2_hip_mem_info": (None, [i32, vp, vp]),
        "rb2_hip_get_counts": (None, [vp, vp]),
        "rb2_hip_rope_bytes": (i64, [vp, i32]),
        "rb2_hip_download_rope": (i64, [vp, i32, vp]),
        "rb2_hip_stream_rope": (i64, [vp, i32, vp, vp]),
        "rb2_hip_load_ropes": (None, [vp, vp, vp]),
        "rb2_hip_rank1a": (None, [vp, i32, i64, vp]),
        "rb2_hip_rank_batch": (None, [vp, i32, i64, vp, vp]),
        "rb2_hip_reserve": (None, [vp, i64, i64, i64]),
        "rb2_hip_num_subropes": (i32, []),
        "rb2_hip_memcpy": (None, [vp, vp, vp, i64, i32]),
        "rb2_hip_use_stream": (None, [vp, vp]),
        "rb2_hip_dev_alloc": (vp, [vp, i64]),
        "rb2_hip_dev_free": (None, [vp, vp]),
        "rb2_hip_synth_reads": (None, [vp, vp, i64, i64, i32, u64, i32]),
        "rb2_hip_synth_reads_cov": (None, [vp, vp, i64, i64, i32, u64, i32, i64]),
        "rb2_hip_synth_reads_skew": (None, [vp, vp, i64, i64, i32, u64, i32, i64, i32]),
        "rb2_hip_sync": (None, [vp]),
        "rb2_hip_sparse_stats": (None, [vp, vp]),
        "rb2_hip_layout_stats": (None, [vp, vp]),
        "rb2_hip_window_stats": (None, [vp, vp]),
        "rb2_hip_host_register": (C.c_int, [vp, C.c_int64]),
        "rb2_hip_host_unregister": (C.c_int, [vp]),
        "rb2_hip_profile": (None, [vp, i32]),
        "rb2_hip_profile_get": (None, [vp, vp, vp, vp, i32]),
        "rb2_hip_kernel_name": (C.c_char_p, [i32]),
        "rb2_hip_layout": (None, [vp, vp, vp]),
        "rb2_hip_multi_create": (vp, [i32, vp, i32, i32, vp]),
        "rb2_hip_multi_unique_id": (None, [vp]),
        "rb2_hip_multi_create_rank": (vp, [i32, i32, i32, vp, i32, vp]),
        "rb2_hip_multi_destroy": (None, [vp]),
        "rb2_hip_default_owners": (None, [i32, vp]),
        "rb2_hip_multi_nranks": (i32, [vp]),
        "rb2_hip_multi_transport": (i32, [vp]),
        "rb2_hip_multi_nlocal": (i32, [vp]),
        "rb2_hip_multi_engine": (vp, [vp, i32]),
        "rb2_hip_multi_insert_multi": (None, [vp, i64, vp]),
        "rb2_hip_multi_insert_multi_dev": (None, [vp, i64, vp]),
        "rb2_hip_multi_get_counts": (None, [vp, vp]),
        "rb2_hip_multi_rope_bytes": (i64, [vp, i32]),
        "rb2_hip_multi_download_rope": (i64, [vp, i32, vp]),
        "rb2_hip_multi_stream_rope": (i64, [vp, i32, vp, vp]),
        "rb2_hip_multi_load_ropes": (None, [vp, vp, vp]),
        "rb2_hip_multi_reserve": (None, [vp, i64, i64, i64]),
        "rb2_hip_multi_reset": (None, [vp]),
        "rb2_hip_multi_sync": (None, [vp]),
        "rb2_hip_multi_rank1a": (None, [vp, i32, i64, vp]),
        "rb2_hip_multi_stats": (None, [vp, vp]),
        "rb2_hip_multi_text_bytes": (i64, [vp, C.c_int]),
        "rb2_hip_multi_rope_hash": (u64, [vp, i32]),
        "rb2_hip_multi_plan_host": (i32, [vp, i32, vp, i32, vp, vp, vp]),
        "rb2_hip_rope_hash": (u64, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            if os.environ.get("RB2_HIP_LIB"):                  # an older build in an A/B run: what it lacks is simply not callable
                continue
            raise
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS = [
    "rb2_hip_device_count", "rb2_hip_set_fatal_handler", "rb2_hip_create", "rb2_hip_destroy", "rb2_hip_sorting_order", "rb2_hip_reset",
    "rb2_hip_insert_multi", "rb2_hip_insert_multi_dev", "rb2_hip_set_lazy", "rb2_hip_wait", "rb2_hip_last_batch_counts", "rb2_hip_prefetch", "rb2_hip_mem_info", "rb2_hip_get_counts", "rb2_hip_rope_bytes",
    "rb2_hip_download_rope", "rb2_hip_stream_rope", "rb2_hip_load_ropes", "rb2_hip_rank1a", "rb2_hip_rank_batch", "rb2_hip_reserve", "rb2_hip_dev_alloc",
    "rb2_hip_num_subropes", "rb2_hip_memcpy", "rb2_hip_use_stream",
    "rb2_hip_dev_free", "rb2_hip_synth_reads", "rb2_hip_synth_reads_cov", "rb2_hip_synth_reads_skew", "rb2_hip_sync", "rb2_hip_sparse_stats", "rb2_hip_layout_stats", "rb2_hip_window_stats", "rb2_hip_host_register", "rb2_hip_host_unregister", "rb2_hip_profile",
    "rb2_hip_profile_get", "rb2_hip_kernel_name", "rb2_hip_layout",
    "rb2_hip_multi_create", "rb2_hip_multi_unique_id", "rb2_hip_multi_create_rank", "rb2_hip_multi_destroy", "rb2_hip_default_owners",
    "rb2_hip_multi_nranks", "rb2_hip_multi_transport", "rb2_hip_multi_nlocal", "rb2_hip_multi_engine", "rb2_hip_multi_insert_multi", "rb2_hip_multi_insert_multi_dev",
    "rb2_hip_multi_get_counts", "rb2_hip_multi_rope_bytes", "rb2_hip_multi_download_rope", "rb2_hip_multi_stream_rope",
    "rb2_hip_multi_load_ropes", "rb2_hip_multi_reserve", "rb2_hip_multi_rope_hash", "rb2_hip_rope_hash", "rb2_hip_multi_plan_host", "rb2_hip_multi_reset", "rb2_hip_multi_sync", "rb2_hip_multi_rank1a", "rb2_hip_multi_stats", "rb2_hip_multi_text_bytes",
]


def expand_runs(rle):
    """1-byte-run 43+3 stream (what the device emits) -> nt6 symbols."""
    rle = np.asarray(rle, dtype=np.uint8)
    return np.repeat(rle & 7, rle >> 3)


def encode_runs(symbols):
    """nt6 symbols -> 43+3 stream using the 1/2/4/8-byte forms (rle.h:53-75); host-side helper."""
    symbols = np.asarray(symbols, dtype=np.uint8)
    out = bytearray()
    if len(symbols) == 0:
        return np.zeros(0, np.uint8)
    edges = np.flatnonzero(np.diff(symbols)) + 1
    starts = np.concatenate([[0], edges])
    lens = np.diff(np.concatenate([starts, [len(symbols)]]))
    for c, l in zip(symbols[starts].tolist(), lens.tolist()):
        if l < 16:
            out.append(l << 3 | c)
        else:
            n = 2 if l < 256 else 4 if l < (1 << 19) else 8
            tail = []
            for _ in range(n - 1):
                tail.append(0x80 | (l & 0x3f))
                l >>= 6
            out.append({2: 0xC0, 4: 0xE0, 8: 0xF0}[n] | l << 3 | c)
            out.extend(reversed(tail))
    return np.frombuffer(bytes(out), dtype=np.uint8)


class HipBwt:
    """Six-rope BWT resident in the HBM of one MI355X."""

    def __init__(self, sorting_order=0, device=0):
        self.L = load_hip_lib()
        if self.L.rb2_hip_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the gfx950 engine has no CPU fallback")
        self.h = self.L.rb2_hip_create(device, sorting_order)
        self.so = sorting_order

    def close(self):
        if getattr(self, "h", None):
            self.L.rb2_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """empty index again; grown buffers are kept"""
        self.L.rb2_hip_reset(self.h)

    # -- the hot path ---------------------------------------------------------------------
    def insert_multi(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.L.rb2_hip_insert_multi(self.h, len(buf), buf.ctypes.data)

    def insert_multi_dev(self, dev_ptr, nbytes):
        self.L.rb2_hip_insert_multi_dev(self.h, nbytes, dev_ptr)

    def set_lazy(self, on):
        """rb2_hip_set_lazy: may insert_multi return before the device is done with the batch (default: yes)"""
        self.L.rb2_hip_set_lazy(self.h, int(on))

    def wait(self):
        self.L.rb2_hip_wait(self.h)

    def last_batch_counts(self):
        """what the last insert_multi adds to counts(), from the text of the batch alone (None if it was not a host-buffer insert)"""
        d = np.zeros(36, np.int64)
        return d.reshape(6, 6) if self.L.rb2_hip_last_batch_counts(self.h, d.ctypes.data) else None

    def prefetch(self, buf, n_final, capacity=0):
        """start uploading buf[:n_final] for a later insert_multi(buf[:len]) (rb2_hip_prefetch); buf must stay alive and in place"""
        self.L.rb2_hip_prefetch(self.h, buf.ctypes.data, n_final, capacity or len(buf))

    # -- state ----------------------------------------------------------------------------
    def counts(self):
        c = np.zeros(36, np.int64)
        self.L.rb2_hip_get_counts(self.h, c.ctypes.data)
        return c.reshape(6, 6)

    def rope_rle(self, b):
        n = self.L.rb2_hip_rope_bytes(self.h, b)
        out = np.zeros(max(n, 1), np.uint8)
        got = self.L.rb2_hip_download_rope(self.h, b, out.ctypes.data)
        assert got == n
        return out[:n]

    def rope(self, b):
        return expand_runs(self.rope_rle(b))

    def ropes(self):
        return [self.rope(b) for b in range(6)]

    def bwt(self):
        return np.concatenate(self.ropes())

    def load_ropes(self, rles):
        arrs = [np.ascontiguousarray(r, dtype=np.uint8) for r in rles]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data if len(a) else None for a in arrs])
        lens = (C.c_int64 * 6)(*[len(a) for a in arrs])
        self.L.rb2_hip_load_ropes(self.h, ptrs, lens)

    def reserve(self, batch_bytes=0, batch_strings=0, total_symbols=0):
        self.L.rb2_hip_reserve(self.h, batch_bytes, batch_strings, total_symbols)

    def rope_hashes(self):
        """device-side checksums of the six ropes (rb2_hip_rope_hash)"""
        return [int(self.L.rb2_hip_rope_hash(self.h, b)) for b in range(6)]

    def rank1a(self, b, x):
        c = np.zeros(6, np.int64)
        self.L.rb2_hip_rank1a(self.h, b, x, c.ctypes.data)
        return c

    def rank_batch(self, b, xs):
        """counts of the six symbols in [0, x) of rope b for every x in xs (one launch, one wave per query)"""
        xs = np.ascontiguousarray(xs, dtype=np.int64)
        out = np.zeros((len(xs), 6), np.int64)
        if len(xs):
            self.L.rb2_hip_rank_batch(self.h, b, len(xs), xs.ctypes.data, out.ctypes.data)
        return out

    # -- measurement helpers ----------------------------------------------------------------
    def dev_alloc(self, nbytes):
        return self.L.rb2_hip_dev_alloc(self.h, nbytes)

    def dev_free(self, p):
        self.L.rb2_hip_dev_free(self.h, p)

    def synth_reads(self, dev_ptr, first, n_reads, read_len, seed=42, strand=0, genome_len=0, skew=0):
        self.L.rb2_hip_synth_reads_skew(self.h, dev_ptr, first, n_reads, read_len, seed, strand, genome_len, skew)

    def sync(self):
        self.L.rb2_hip_sync(self.h)

    def sparse_stats(self):
        a = np.zeros(4, np.int64)
        self.L.rb2_hip_sparse_stats(self.h, a.ctypes.data)
        return {"relayouts": int(a[0]), "void_rounds": int(a[1]), "sparse_rounds": int(a[2]), "sparse_now": bool(a[3])}

    def layout_stats(self):
        a = np.zeros(8, np.int64)
        self.L.rb2_hip_layout_stats(self.h, a.ctypes.data)
        return {"relayouts": int(a[0]), "void_rounds": int(a[1]), "sparse_rounds": int(a[2]), "sparse_now": bool(a[3]),
                "respreads": int(a[4]), "leaf_splits": int(a[5]), "grown_in_rounds": int(a[6]), "plain_handovers": int(a[7])}

    def window_stats(self):
        a = np.zeros(6, np.int64)
        self.L.rb2_hip_window_stats(self.h, a.ctypes.data)
        return {"plain": int(a[0]), "compact0": int(a[1]), "compact1": int(a[2]), "compact2": int(a[3]), "compact_rounds": int(a[4]), "counted": bool(a[5])}

    def profile(self, on=True):
        """False / 0: off; True / 1: every kernel group of a round; 2: the merge launches only (two events per round instead of sixteen)"""
        self.L.rb2_hip_profile(self.h, int(on))

    def profile_get(self, reset=False):
        n = len(K_NAMES)
        la = np.zeros(n, np.int64)
        ms = np.zeros(n, np.float64)
        un = np.zeros(n, np.int64)
        self.L.rb2_hip_profile_get(self.h, la.ctypes.data, ms.ctypes.data, un.ctypes.data, 1 if reset else 0)
        return {K_NAMES[i]: {"launches": int(la[i]), "ms": float(ms[i]), "units": int(un[i])} for i in range(n)}

    @staticmethod
    def layout():
        L = load_hip_lib()
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.rb2_hip_layout(C.byref(a), C.byref(b), C.byref(c))
        return {"leaf_syms": a.value, "tile_leaves": b.value, "string_tile": c.value}


TRANSPORTS = {"peer": 0, "rccl": 1}


class _Engine(HipBwt):
    """a local rank's engine inside a MultiBwt (borrowed handle: never destroyed from here)"""

    def __init__(self, lib, h, so):
        self.L, self.h, self.so = lib, h, so

    def close(self):
        self.h = None


class MultiBwt:
    """One BWT whose 31 sub-ropes are sharded over N ranks, driven inside the library (include/rb2_hip.h, rb2_hip_multi_*):
    the same ``insert_multi(buf)`` contract as ``HipBwt`` / mr_insert_multi (mrope.c:258), no Python in the round loop.

    ``devices``: one entry per rank; the same device may appear several times (virtual ranks on one GPU, PEER transport).
    ``rank``/``nranks``/``nccl_id``: this process is ONE rank of a multi-process group over RCCL (bench.py under torchrun)."""

    def __init__(self, sorting_order, devices, transport="peer", owners=None, rank=None, nranks=None, nccl_id=None):
        self.L = load_hip_lib()
        if self.L.rb2_hip_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the gfx950 engine has no CPU fallback")
        own = (C.c_int * len(owners))(*owners) if owners is not None else None
        self.so = sorting_order
        if rank is None:
            devs = (C.c_int * len(devices))(*devices)
            self.h = self.L.rb2_hip_multi_create(len(devices), devs, sorting_order, TRANSPORTS[transport], own)
        else:
            idb = C.create_string_buffer(bytes(nccl_id), 128) if nccl_id is not None else None
            self.h = self.L.rb2_hip_multi_create_rank(devices[0], rank, nranks, idb, sorting_order, own)
        self.n = self.L.rb2_hip_multi_nlocal(self.h)
        self.world = self.L.rb2_hip_multi_nranks(self.h)

    @staticmethod
    def unique_id():
        b = C.create_string_buffer(128)
        load_hip_lib().rb2_hip_multi_unique_id(b)
        return b.raw

    @staticmethod
    def default_owners(nranks):
        a = (C.c_int * 31)()
        load_hip_lib().rb2_hip_default_owners(nranks, a)
        return list(a)

    def text_bytes(self):
        """bytes of device memory every local rank holds for the text of the last host-buffer batch (rb2_hip_multi_text_bytes)"""
        return [int(self.L.rb2_hip_multi_text_bytes(self.h, k)) for k in range(self.n)]

    def close(self):
        if getattr(self, "h", None):
            self.L.rb2_hip_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def engine(self, k=0):
        return _Engine(self.L, self.L.rb2_hip_multi_engine(self.h, k), self.so)

    def insert_multi(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.L.rb2_hip_multi_insert_multi(self.h, len(buf), buf.ctypes.data)

    def insert_multi_dev(self, dev_ptrs, nbytes):
        """dev_ptrs: one device pointer per local rank (or a single pointer all local ranks share)"""
        if not isinstance(dev_ptrs, (list, tuple)):
            dev_ptrs = [dev_ptrs] * self.n
        arr = (C.c_void_p * self.n)(*dev_ptrs)
        self.L.rb2_hip_multi_insert_multi_dev(self.h, nbytes, arr)

    def counts(self):
        c = np.zeros(36, np.int64)
        self.L.rb2_hip_multi_get_counts(self.h, c.ctypes.data)
        return c.reshape(6, 6)

    def rope_rle(self, b):
        n = self.L.rb2_hip_multi_rope_bytes(self.h, b)
        out = np.zeros(max(n, 1), np.uint8)
        got = self.L.rb2_hip_multi_download_rope(self.h, b, out.ctypes.data)
        assert got == n
        return out[:n]

    def rope(self, b):
        return expand_runs(self.rope_rle(b))

    def load_ropes(self, rles):
        arrs = [np.ascontiguousarray(r, dtype=np.uint8) for r in rles]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data if len(a) else None for a in arrs])
        lens = (C.c_int64 * 6)(*[len(a) for a in arrs])
        self.L.rb2_hip_multi_load_ropes(self.h, ptrs, lens)

    def reserve(self, batch_bytes=0, batch_strings=0, total_symbols=0):
        self.L.rb2_hip_multi_reserve(self.h, batch_bytes, batch_strings, total_symbols)

    def reset(self):
        self.L.rb2_hip_multi_reset(self.h)

    def rope_hashes(self):
        return [int(self.L.rb2_hip_multi_rope_hash(self.h, b)) for b in range(6)]

    def sync(self):
        self.L.rb2_hip_multi_sync(self.h)

    def rank1a(self, b, x):
        c = np.zeros(6, np.int64)
        self.L.rb2_hip_multi_rank1a(self.h, b, x, c.ctypes.data)
        return c

    def stats(self):
        a = np.zeros(6, np.int64)
        self.L.rb2_hip_multi_stats(self.h, a.ctypes.data)
        return {"host_syncs_in_rounds": int(a[0]), "rounds": int(a[1]), "batches": int(a[2]), "sparse_rounds": int(a[3]),
                "void_rounds": int(a[4]), "relayouts": int(a[5])}
