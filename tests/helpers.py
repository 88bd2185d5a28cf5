"""Shared test plumbing: synthetic reads, nt6 batch encoding, oracle/_ref access.

Everything under oracle/ is the CHECKER.  The product (ropebwt2_amd) never imports this file.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "ropebwt2")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libropebwt2_ref.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GEN = os.path.join(ROOT, "ropebwt2_amd", "bin", "synth_reads")      # tools/synth_reads.c, built by ropebwt2_amd/build.py

NT6 = np.full(256, 5, dtype=np.uint8)          # seq_nt6_table, main.c:17-26
NT6[0] = 0
for _ch, _v in (("A", 1), ("C", 2), ("G", 3), ("T", 4)):
    NT6[ord(_ch)] = _v
    NT6[ord(_ch.lower())] = _v
SYMS = np.frombuffer(b"$ACGTN", dtype=np.uint8)
MASK64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------
# synthetic reads: base j of read i = "ACGT"[splitmix64(seed, i*L+j+1) >> 62]
# ---------------------------------------------------------------------------------------------

def splitmix_bases(n_reads, read_len, seed=42, first=0):
    """uint8 array [n_reads, read_len] of nt6 codes 1..4."""
    k = (np.arange(first * read_len, (first + n_reads) * read_len, dtype=np.uint64) + np.uint64(1))
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(62)).astype(np.uint8) + 1).reshape(n_reads, read_len)


def reads_to_text(codes):
    """[n, L] nt6 codes -> bytes of one-read-per-line text."""
    n, L = codes.shape
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = SYMS[codes]
    out[:, L] = 10
    return out.tobytes()


def repetitive_reads(n_reads, seed=7, genome_len=150, max_len=40, p_n=0.05):
    """Highly repetitive, variable-length (0..max_len) reads sampled from a short genome, with a
    few N-containing reads and exact duplicates: exercises rank2a on non-empty intervals, empty
    strings and all six ropes (fixture class (ii))."""
    rng = np.random.RandomState(seed)
    genome = rng.randint(1, 5, size=genome_len).astype(np.uint8)
    out = []
    for _ in range(n_reads):
        ln = rng.randint(0, max_len + 1)
        st = rng.randint(0, genome_len - ln + 1)
        r = genome[st:st + ln].copy()
        if ln and rng.rand() < p_n:
            r[rng.randint(0, ln)] = 5
        out.append(r)
    return out


def lines_from_codes(reads):
    return b"".join(SYMS[np.asarray(r, dtype=np.uint8)].tobytes() + b"\n" for r in reads)


# ---------------------------------------------------------------------------------------------
# batch buffer = what main.c hands to mr_insert_multi (main.c:200-237): every read reversed,
# 0-terminated; with both strands the reverse strand (complement of the original, unreversed)
# follows its forward strand.
# ---------------------------------------------------------------------------------------------

def encode_batch(reads, fwd=True, rev=False):
    parts = []
    for r in reads:
        r = np.asarray(r, dtype=np.uint8)
        if fwd:
            parts.append(r[::-1])
            parts.append(np.zeros(1, np.uint8))
        if rev:
            c = r.copy()
            m = (c >= 1) & (c <= 4)
            c[m] = 5 - c[m]
            parts.append(c)
            parts.append(np.zeros(1, np.uint8))
    if not parts:
        return np.zeros(0, np.uint8)
    return np.ascontiguousarray(np.concatenate(parts))


def encode_batch_fixed(codes, fwd=True, rev=False):
    """Vectorised encode_batch for an [n, L] array (fixed-length reads)."""
    n, L = codes.shape
    rows = []
    if fwd:
        f = np.zeros((n, L + 1), np.uint8)
        f[:, :L] = codes[:, ::-1]
        rows.append(f)
    if rev:
        r = np.zeros((n, L + 1), np.uint8)
        c = codes.copy()
        m = (c >= 1) & (c <= 4)
        c[m] = 5 - c[m]
        r[:, :L] = c
        rows.append(r)
    if len(rows) == 1:
        return np.ascontiguousarray(rows[0].reshape(-1))
    return np.ascontiguousarray(np.stack(rows, axis=1).reshape(-1))


def text_to_reads(text):
    """one-read-per-line text -> list of nt6 arrays (main.c -L path: stop at first non-alpha)."""
    out = []
    for ln in text.split(b"\n")[:-1] if text.endswith(b"\n") else text.split(b"\n"):
        a = np.frombuffer(ln, dtype=np.uint8)
        alpha = ((a >= 65) & (a <= 90)) | ((a >= 97) & (a <= 122))
        stop = len(a) if alpha.all() else int(np.argmin(alpha))
        out.append(NT6[a[:stop]])
    return out


# ---------------------------------------------------------------------------------------------
# oracle (plain-C restatement) through ctypes
# ---------------------------------------------------------------------------------------------

_oracle = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in (L.orc_insert_multi, L.orc_insert_multi_seq):
            f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            f.restype = None
        L.orc_insert1.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_rope_len.argtypes = [C.c_void_p, C.c_int]
        L.orc_rope_len.restype = C.c_int64
        L.orc_rope_ptr.argtypes = [C.c_void_p, C.c_int]
        L.orc_rope_ptr.restype = C.c_void_p
        L.orc_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_total.argtypes = [C.c_void_p]
        L.orc_total.restype = C.c_int64
        L.orc_bwt.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_bwt.restype = C.c_int64
        L.orc_rank1a.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_rle_enc1.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        L.orc_rle_dec1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _oracle = L
    return _oracle


class Oracle:
    """BWT of a growing collection, built by the plain-C restatement."""

    def __init__(self, so):
        self.L = oracle_lib()
        self.h = self.L.orc_create(so)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    __del__ = close

    def insert_multi(self, buf, seq=False):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        fn = self.L.orc_insert_multi_seq if seq else self.L.orc_insert_multi
        fn(self.h, len(buf), buf.ctypes.data)

    def insert1(self, rev_str):
        s = np.ascontiguousarray(np.concatenate([np.asarray(rev_str, np.uint8), np.zeros(1, np.uint8)]))
        self.L.orc_insert1(self.h, s.ctypes.data)

    def rope(self, b):
        n = self.L.orc_rope_len(self.h, b)
        if n == 0:
            return np.zeros(0, np.uint8)
        p = self.L.orc_rope_ptr(self.h, b)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def ropes(self):
        return [self.rope(b) for b in range(6)]

    def bwt(self):
        return np.concatenate(self.ropes())

    def counts(self):
        c = np.zeros(36, np.int64)
        self.L.orc_counts(self.h, c.ctypes.data)
        return c.reshape(6, 6)


def decode_runs(rle):
    """43+3 run bytes (any run width, rle.h:39-51) -> nt6 symbols; small inputs only (Python loop)"""
    rle = np.asarray(rle, dtype=np.uint8)
    out, i, n = [], 0, len(rle)
    while i < n:
        b = int(rle[i]); c = b & 7
        if b & 0x80 == 0:
            l = b >> 3; i += 1
        elif b >> 5 == 6:
            l = (b & 0x18) << 3 | (int(rle[i + 1]) & 0x3f); i += 2
        else:
            nb = 8 if b & 0x10 else 4
            l = (b >> 3) & 1
            for k in range(1, nb):
                l = l << 6 | (int(rle[i + k]) & 0x3f)
            i += nb
        out.append(np.full(l, c, np.uint8))
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


def bwt_text(codes):
    return SYMS[np.asarray(codes, dtype=np.uint8)].tobytes()


# ---------------------------------------------------------------------------------------------
# the real reference, when its build products are present (oracle/_ref)
# ---------------------------------------------------------------------------------------------

def have_ref():
    return os.path.exists(REF_BIN)


def run_ref(flags, text, extra=()):
    """Run the reference CLI on one-read-per-line text; returns stdout bytes."""
    p = subprocess.run([REF_BIN] + list(flags) + list(extra) + ["-"], input=text,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    return p.stdout


def md5(b):
    return hashlib.md5(b).hexdigest()


# ---- sub-ropes: Python restatement of the owner map and the exchange layout of the sharded build (rb2_device.h rope_of /
# ---- rope_sym / rope_prev, rb2_hip_default_owners, shard_layout in rb2_engine.hip) for the tests that check the library's own

NR = 31                 # sub-ropes: rope $ + pieces (b,x)
REC_WORDS = 3           # a string record on the wire: 24 bytes (rb2_device.h ShardRec)


def rope_sym(r):
    """rope the piece belongs to (0 = $ ... 5 = N)"""
    return 0 if r == 0 else (r - 1) // 6 + 1


def rope_prev(r):
    """x of piece (b,x): the symbol following b in the suffix"""
    return 0 if r == 0 else (r - 1) % 6


def rope_of(b, x):
    return 0 if b == 0 else 1 + (b - 1) * 6 + x


def default_owners(nranks):
    """piece -> rank.  On DNA the 16 pieces (b,x), b,x in ACGT, carry ~1/16 of the rows each; they are dealt out in contiguous
    blocks, so up to 16 ranks get load.  The light pieces ((b,$): one row per read; everything with N) ride with a neighbour,
    rope $ with rank 0."""
    own = [0] * NR
    if nranks <= 1:
        return own
    for b in range(1, 6):
        for x in range(6):
            k = (min(b, 4) - 1) * 4 + (min(max(x, 1), 4) - 1)
            own[rope_of(b, x)] = min(k * nranks // 16, nranks - 1)
    return own


def exchange_layout(owner, nranks, src, g):
    """records rank ``src`` sends to every rank for the count matrix g (NR x 6): the members of bucket r that insert a
    (a = 1..5) travel from owner[r] to owner[(a, rope_sym(r))]"""
    g = np.asarray(g, dtype=np.int64).reshape(NR, 6)
    per = [0] * nranks
    for r in range(NR):
        if owner[r] != src:
            continue
        for a in range(1, 6):
            per[owner[rope_of(a, rope_sym(r))]] += int(g[r, a])
    return per


def exchange_block(owner, src, dst, g):
    """the records src sends to dst in the order they sit in src's send buffer, tagged (r, a, index): for the pieces r2 = (a,b)
    owned by dst (ascending), for the pieces r of rope b owned by src (ascending), the g[r][a] members in their old order"""
    out = []
    for r2 in range(1, NR):
        if owner[r2] != dst:
            continue
        a, b = rope_sym(r2), rope_prev(r2)
        for r in range(NR):
            if rope_sym(r) == b and owner[r] == src:
                out += [(r, a, i) for i in range(int(g[r, a]))]
    return out
