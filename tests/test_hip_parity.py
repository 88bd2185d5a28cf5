"""GPU parity tests: the HIP hot path (through the C ABI of include/rb2_hip.h) must return exactly
the BWT the oracle / the reference returns -- bit-exact, every rope, every count.

Structure mirrors how one would test mr_insert_multi itself: build the batch buffer main.c would
build (helpers.encode_batch*), call insert_multi, compare the six ropes.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
SO_FLAG = {0: "-LR", 1: "-LRs", 2: "-LRr"}


def run_both(hip, so, batches):
    o = H.Oracle(so)
    g = hip.HipBwt(so)
    for i, buf in enumerate(batches):
        o.insert_multi(buf)
        g.insert_multi(buf)
        assert np.array_equal(o.counts(), g.counts()), "count matrix differs after batch %d" % i
    for b in range(6):
        ro, rg = o.rope(b), g.rope(b)
        assert len(ro) == len(rg), "rope %d length" % b
        assert np.array_equal(ro, rg), "rope %d differs at %s" % (b, np.flatnonzero(ro != rg)[:5])
    return o, g


@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("both", [False, True])
def test_kat(hip, golden, so, both):
    reads = H.text_to_reads(golden["kat_input"].encode())
    o, g = run_both(hip, so, [H.encode_batch(reads, True, both)])
    flag = SO_FLAG[so] if not both else SO_FLAG[so].replace("R", "")
    assert H.bwt_text(g.bwt()).decode() == golden["kat"][flag]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_repetitive_multibatch(hip, so):
    """variable length 0..40, duplicates, N's, empty strings, three batches, last with both strands:
    rank2a on non-empty intervals in every round (mrope.c:199-202)."""
    reads = H.repetitive_reads(1500, seed=11 + so)
    run_both(hip, so, [H.encode_batch(reads[:500]), H.encode_batch(reads[500:1000]), H.encode_batch(reads[1000:], True, True)])


@pytest.mark.parametrize("so", [0, 1, 2])
def test_repetitive_larger(hip, so):
    reads = H.repetitive_reads(20000, seed=5 + so, genome_len=3000, max_len=120)
    run_both(hip, so, [H.encode_batch(reads[:10000]), H.encode_batch(reads[10000:])])


@pytest.mark.parametrize("so", [0, 1, 2])
def test_random_10k_golden(hip, golden, so):
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    _, dev = run_both(hip, so, [H.encode_batch_fixed(codes[:5000]), H.encode_batch_fixed(codes[5000:])])
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


def test_random_10k_both_strands_golden(hip, golden):
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    dev = hip.HipBwt(2)
    dev.insert_multi(H.encode_batch_fixed(codes, True, True))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"]["-Lr"]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_long_reads(hip, golden, so):
    """config 4 shape (long strings, few per round): 200 x 10 kbp, golden from the reference."""
    g = golden["sets"]["200_x_10k"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    dev = hip.HipBwt(so)
    dev.insert_multi(H.encode_batch_fixed(codes))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_golden_100k(hip, golden, so):
    g = golden["sets"]["100k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    dev = hip.HipBwt(so)
    for i in range(0, 100000, 40000):          # uneven batches
        dev.insert_multi(H.encode_batch_fixed(codes[i:i + 40000]))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [1, 2])
def test_golden_1M(hip, golden, so):
    g = golden["sets"]["1M_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    dev = hip.HipBwt(so)
    dev.insert_multi(H.encode_batch_fixed(codes[:600000]))
    dev.insert_multi(H.encode_batch_fixed(codes[600000:]))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


# ---- edge cases --------------------------------------------------------------------------------

@pytest.mark.parametrize("so", [0, 1, 2])
def test_empty_and_tiny_strings(hip, so):
    batches = [np.zeros(1, np.uint8),                                  # one empty string
               np.zeros(7, np.uint8),                                  # only empty strings
               np.array([3, 0], np.uint8),                             # one 1-symbol string
               H.encode_batch([[1], [], [1], [5], [4, 4], []])]
    run_both(hip, so, batches)


@pytest.mark.parametrize("so", [0, 1, 2])
def test_homopolymers_long_runs(hip, so):
    """runs far longer than 15 (the 1-byte run limit of the device leaves) and than one leaf"""
    lay = hip.HipBwt.layout()
    n = lay["leaf_syms"] * lay["tile_leaves"] + 37
    reads = [[1] * 70] * 300 + [[4] * 33] * 200 + [[2] * 5 + [5] * 3] * 100 + [[3] * (n // 64)] * 70
    run_both(hip, so, [H.encode_batch(reads[:400]), H.encode_batch(reads[400:])])


@pytest.mark.parametrize("so", [0, 1])
@pytest.mark.parametrize("delta", [-1, 0, 1])
def test_leaf_and_tile_boundaries(hip, so, delta):
    """rope sizes that hit leaf / merge-tile / string-tile multiples exactly and off by one"""
    lay = hip.HipBwt.layout()
    n = lay["string_tile"] * 8 + delta             # strings: multiple of the string tile (+-1)
    codes = H.splitmix_bases(n, 7, seed=3)
    run_both(hip, so, [H.encode_batch_fixed(codes)])
    T = lay["leaf_syms"] * lay["tile_leaves"]
    one = np.concatenate([np.full(T + delta - 1, 2, np.uint8), np.zeros(1, np.uint8)])   # single string filling one tile
    run_both(hip, so, [one, one])


def test_reserve_is_only_a_hint(hip):
    """rb2_hip_reserve before, between and after batches (smaller and larger than needed) never changes the result"""
    codes = H.splitmix_bases(6000, 40, seed=12)
    bufs = [H.encode_batch_fixed(codes[:2500]), H.encode_batch_fixed(codes[2500:])]
    o = H.Oracle(1)
    dev = hip.HipBwt(1)
    dev.reserve(10, 1, 100)                                 # far too small: grows on demand
    for i, buf in enumerate(bufs):
        o.insert_multi(buf); dev.insert_multi(buf)
        dev.reserve(len(buf) * 3, 20000, 5_000_000)         # larger than needed, index already populated
    assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b))


@pytest.mark.parametrize("reserve", [False, True])
def test_no_buffer_grows_while_rounds_are_queued(hip, reserve):
    """everything a dense batch needs is sized before its first round: a buffer that grows in the middle of the queued rounds is a
    hipFree / a new mapping, i.e. a device-wide wait the host thread sits in (rb2_hip_layout_stats out[6]; r05: the superblock totals
    grew round by round).  Three host-buffer batches on a growing index, with and without the capacity hint."""
    codes = H.splitmix_bases(900_000, 101, seed=77)
    bufs = [H.encode_batch_fixed(codes[i:i + 300_000]) for i in (0, 300_000, 600_000)]
    old = os.environ.get("RB2_SPARSE_LAMBDA")
    os.environ["RB2_SPARSE_LAMBDA"] = "0"                   # (dense rounds only: re-layouts size their pools themselves)
    try:
        dev = hip.HipBwt(1)
    finally:
        if old is None: os.environ.pop("RB2_SPARSE_LAMBDA", None)
        else: os.environ["RB2_SPARSE_LAMBDA"] = old
    if reserve:
        dev.reserve(len(bufs[0]), 300_000, 3 * len(bufs[0]))
    for buf in bufs:
        dev.insert_multi(buf)
    st = dev.layout_stats()
    assert int(dev.counts().sum()) == sum(len(b) for b in bufs)
    dev.close()
    assert st["grown_in_rounds"] == 0, st


@pytest.mark.parametrize("fold", [None, 0])
@pytest.mark.parametrize("so", [0, 1, 2])
def test_tile_scan_over_several_chunks(hip, so, fold):
    """1.2 M reads of 10 bp are 2344 string tiles = three chunks of the tile scan (k_tscan1 / k_tscan3, the many-tiles path forced by
    RB2_TS_MAX): the chunk in the middle needs the totals in front of it and the next group head behind it; 4^10 possible reads make
    groups that span tiles and chunks in the sorted orders.  With the scan over the chunk totals folded into k_tscan3 (default) and as
    a launch of its own (RB2_TS_FOLD=0, what batches of more than 2^20 tiles get)."""
    rng = np.random.default_rng(17 + so)
    a = rng.integers(1, 5, size=(700_000, 10), dtype=np.uint8)
    b = rng.integers(1, 5, size=(500_000, 10), dtype=np.uint8)
    knobs = {"RB2_TS_MAX": "2"}
    if fold is not None: knobs["RB2_TS_FOLD"] = str(fold)
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        dev, o = hip.HipBwt(so), H.Oracle(so)
        for codes in (a, b):
            buf = H.encode_batch_fixed(codes)
            o.insert_multi(buf); dev.insert_multi(buf)
        assert np.array_equal(dev.counts(), o.counts())
        for r in range(6):
            assert np.array_equal(dev.rope(r), o.rope(r)), "rope %d" % r
        dev.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


@pytest.mark.parametrize("so", [0, 1])
def test_sparse_inserts_into_large_index(hip, so):
    """steady state of a long job: a small batch into an index hundreds of merge windows long (few new symbols
    per window: the gap-opening path), then a batch with long homopolymers (many per word: the dense path)"""
    base = H.splitmix_bases(30000, 101, seed=5)
    small = H.splitmix_bases(700, 101, seed=6)
    dense = [[1] * 90] * 400 + [[2, 2, 2, 3] * 20] * 300
    run_both(hip, so, [H.encode_batch_fixed(base), H.encode_batch_fixed(small), H.encode_batch(dense)])


@pytest.mark.parametrize("so", [0, 1, 2])
def test_fuzz_small_shapes(hip, so):
    """many random small jobs: string counts around the tile / window / leaf sizes, lengths 0..80 with a heavy tail,
    N's, duplicates, 1-3 batches, both strands now and then; every rope compared with the oracle"""
    rng = np.random.RandomState(1000 + so + 7919 * int(os.environ.get("RB2_FUZZ_SEED", "0")))
    for it in range(int(os.environ.get("RB2_FUZZ_ITERS", "24"))):          # soak runs: RB2_FUZZ_ITERS=400 RB2_FUZZ_SEED=k
        nb = rng.randint(1, 4)
        batches = []
        for _ in range(nb):
            n = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 511, 512, 513, 1500]))
            pool = [list(rng.randint(1, 5, size=rng.randint(1, 30))) for _ in range(5)]       # shared substrings -> equal suffixes
            reads = []
            for _ in range(n):
                L = int(rng.choice([0, 1, 2, 5, 17, 40, 80], p=[.05, .1, .1, .2, .25, .2, .1]))
                if rng.rand() < 0.3:
                    r = (pool[rng.randint(5)] * 4)[:L]
                else:
                    r = list(rng.randint(1, 5, size=L))
                if len(r) and rng.rand() < 0.1:
                    r[rng.randint(len(r))] = 5
                reads.append([int(x) for x in r])
            both = rng.rand() < 0.25
            batches.append(H.encode_batch(reads, True, True) if both else H.encode_batch(reads))
        run_both(hip, so, batches)


@pytest.mark.parametrize("so,strand", [(0, 0), (1, 0), (2, 1)])
def test_coverage_reads_match_oracle(hip, so, strand):
    """reads cut from one random genome at ~60x coverage (rb2_hip_synth_reads_cov): every round has large groups and
    non-empty intervals -- the regime of real sequencing data, which i.i.d. reads leave after ~14 rounds.  Two batches."""
    n, L, glen = 12000, 50, 10000
    per = (L + 1) * (2 if strand else 1)
    o = H.Oracle(so)
    dev = hip.HipBwt(so)
    p = dev.dev_alloc(n * per + 64)
    for first in (0, n):
        dev.synth_reads(p, first, n, L, seed=11, strand=strand, genome_len=glen)
        dev.sync()
        host = np.empty(n * per, np.uint8)
        dev.L.rb2_hip_memcpy(dev.h, host.ctypes.data, p, n * per, 1)
        assert host[-1] == 0 and int((host == 0).sum()) == n * (2 if strand else 1)
        dev.insert_multi_dev(p, n * per)
        o.insert_multi(host)
        assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b)), "rope %d" % b
    # the reads really overlap: far fewer distinct 20-mers than i.i.d. reads would have
    kmers = {bytes(host[i * per + 5:i * per + 25]) for i in range(0, n, 7)}
    assert len(kmers) < 0.95 * len(range(0, n, 7))
    dev.dev_free(p)


@pytest.mark.parametrize("badval,pos", [(6, 5), (7, 70), (200, 3000), (9, -2)])
def test_invalid_symbols_fail_loudly(hip, badval, pos):
    """a byte outside 0..5 would corrupt the bucket bookkeeping: the engine refuses the batch (in a child process: abort)"""
    import subprocess, sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from ropebwt2_amd import HipBwt; "
            "b = np.tile(np.array([1, 2, 3, 4, 0], np.uint8), 1000); b[%d] = %d; HipBwt(1).insert_multi(b); print('inserted')"
            % (H.ROOT, pos, badval))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"inserted" not in p.stdout
    assert b"not nt6 codes" in p.stderr


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_fuzz_medium_jobs(hip, seed):
    """medium-size random jobs (0.1-0.3 M reads of 30-150 bp in 2-4 batches, i.i.d. or cut from a genome at 5-80x, random
    order / strands): long enough for many merge windows per sub-rope, tile-spanning groups and both interval regimes"""
    rng = np.random.RandomState(500 + seed)
    so, strand = int(rng.randint(3)), int(rng.randint(2))
    L = int(rng.randint(30, 151))
    nb = int(rng.randint(2, 5))
    n = int(rng.randint(100000, 300001)) // nb
    cov = float(rng.choice([0, 5, 20, 80]))
    glen = int(max(L + 1, n * nb * L / cov)) if cov else 0
    per = (L + 1) * (2 if strand else 1)
    o, dev = H.Oracle(so), hip.HipBwt(so)
    p = dev.dev_alloc(n * per + 64)
    for b in range(nb):
        dev.synth_reads(p, b * n, n, L, seed=100 + seed, strand=strand, genome_len=glen)
        dev.sync()
        host = np.empty(n * per, np.uint8)
        dev.L.rb2_hip_memcpy(dev.h, host.ctypes.data, p, n * per, 1)
        dev.insert_multi_dev(p, n * per)
        o.insert_multi(host)
        assert np.array_equal(dev.counts(), o.counts()), "counts after batch %d (so %d strand %d L %d cov %g)" % (b, so, strand, L, cov)
    for r in range(6):
        assert np.array_equal(dev.rope(r), o.rope(r)), "rope %d (so %d strand %d L %d cov %g)" % (r, so, strand, L, cov)
    dev.dev_free(p)


@pytest.mark.parametrize("so", [0, 1])
def test_one_symbol_fills_several_superblocks(hip, so):
    """a sub-rope that is one symbol over more than two superblocks (2 x 32 leaves): per-symbol counts reach the leaf and
    superblock maxima in the packed 16-bit scans of k_merge / k_meta_sb; the export emits maximal runs"""
    lay = hip.HipBwt.layout()
    n_sym = lay["leaf_syms"] * 32 * 2 + 5000                         # > 2 superblocks of A's in piece (A,A)
    reads = [[1] * 100] * (n_sym // 100 + 1) + [[2] * 7, [1, 2, 1, 2]]
    run_both(hip, so, [H.encode_batch(reads[:len(reads) // 2]), H.encode_batch(reads[len(reads) // 2:])])


def test_single_long_string(hip):
    codes = H.splitmix_bases(1, 50000, seed=77)
    for so in (0, 1):
        run_both(hip, so, [H.encode_batch_fixed(codes)])


@pytest.mark.parametrize("so", [0, 1, 2])
def test_incremental_on_loaded_index(hip, so):
    """config 5 shape: seed the device from an existing index (what mr_restore would hand over,
    with 2/4/8-byte runs in the stream), then insert more; equals the one-shot build."""
    from ropebwt2_amd.hipbwt import encode_runs
    reads = H.repetitive_reads(3000, seed=50 + so, genome_len=400, max_len=60) + [[1] * 300] * 40
    o = H.Oracle(so)
    o.insert_multi(H.encode_batch(reads[:1500]))
    dev = hip.HipBwt(so)
    dev.load_ropes([encode_runs(o.rope(b)) for b in range(6)])
    assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b))
    buf = H.encode_batch(reads[1500:])
    o.insert_multi(buf)
    dev.insert_multi(buf)
    assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b))


def test_load_index_with_very_long_runs(hip):
    """the device-side loader (k_ld_*): runs of tens of thousands of symbols (4-byte codes, spread over many words and leaves by
    k_ld_long), runs cut by piece and leaf borders, then more inserts on top; ropes equal the oracle's before and after"""
    from ropebwt2_amd.hipbwt import encode_runs
    reads = [[1] * 3000] * 20 + [[2] * 2500] * 12 + [[1, 2] * 700] * 5 + H.repetitive_reads(500, seed=9, genome_len=300, max_len=50)
    for so in (0, 1):
        o = H.Oracle(so)
        o.insert_multi(H.encode_batch(reads))
        dev = hip.HipBwt(so)
        dev.load_ropes([encode_runs(o.rope(b)) for b in range(6)])
        assert np.array_equal(dev.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(dev.rope(b), o.rope(b)), "rope %d after load (so %d)" % (b, so)
        more = H.encode_batch([[1] * 900, [3, 1, 1, 1, 2] * 40] + H.repetitive_reads(200, seed=10, genome_len=300, max_len=50))
        o.insert_multi(more); dev.insert_multi(more)
        for b in range(6):
            assert np.array_equal(dev.rope(b), o.rope(b)), "rope %d after insert (so %d)" % (b, so)
        dev.close()


def test_rank1a_matches_oracle(hip):
    reads = H.repetitive_reads(4000, seed=3, genome_len=500, max_len=80)
    o, dev = run_both(hip, 1, [H.encode_batch(reads)])
    rng = np.random.RandomState(0)
    for b in range(6):
        r = o.rope(b)
        for x in [0, 1, len(r) // 2, len(r) - 1, len(r)] + list(rng.randint(0, len(r) + 1, size=5)):
            x = int(max(0, min(len(r), x)))
            assert np.array_equal(dev.rank1a(b, x), np.bincount(r[:x], minlength=6))


@pytest.mark.parametrize("strand", [0, 1])
def test_device_generator_matches_host(hip, strand):
    """the on-device synthetic reads (bench.py input) are the SURVEY.md 8c stream, encoded as main.c would"""
    n, L = 3000, 37
    codes = H.splitmix_bases(n, L, seed=42, first=100)
    a = hip.HipBwt(2)
    a.insert_multi(H.encode_batch_fixed(codes, True, bool(strand)))
    b = hip.HipBwt(2)
    nbytes = n * (L + 1) * (2 if strand else 1)
    p = b.dev_alloc(nbytes)
    b.synth_reads(p, 100, n, L, seed=42, strand=strand)
    b.insert_multi_dev(p, nbytes)
    b.dev_free(p)
    assert np.array_equal(a.counts(), b.counts())
    for r in range(6):
        assert np.array_equal(a.rope_rle(r), b.rope_rle(r))


# ---- full batch size: size-independent properties ------------------------------------------------

def _sym_at(dev, b, p):
    return int(np.argmax(dev.rank1a(b, p + 1) - dev.rank1a(b, p)))


def _walk(dev, counts, row):
    """spell the string whose '$'-suffix sits at row `row` of rope $ by LF-mapping (inverse BWT)."""
    out = []
    b, p = 0, row
    while True:
        c = _sym_at(dev, b, p)
        if c == 0:
            return out
        out.append(c)
        p = int(counts[:b, c].sum() + dev.rank1a(b, p)[c])
        b = c


def test_full_batch_properties(hip):
    """one full -m4g batch (40.8 M x 101 bp, BASELINE.json configs[1]) in input order:
    (1) every rope's size equals the number of occurrences of its symbol (LF consistency),
    (2) inverse-BWT walks from sampled rows of rope $ reproduce exactly the sampled reads."""
    L = 101
    n = -(-(int(4 * 1024 ** 3 * 0.97) + 1) // (L + 1))
    dev = hip.HipBwt(0)
    p = dev.dev_alloc(n * (L + 1))
    dev.synth_reads(p, 0, n, L, seed=42)
    dev.insert_multi_dev(p, n * (L + 1))
    dev.dev_free(p)
    c = dev.counts()
    assert c.sum() == n * (L + 1)
    assert c[:, 0].sum() == n and c[0].sum() == n
    for b in range(1, 6):
        assert c[b].sum() == c[:, b].sum()
    for k in [0, 1, 12345, n // 2, n - 1]:
        got = _walk(dev, c, k)                      # read k reversed
        want = H.splitmix_bases(1, L, seed=42, first=k)[0][::-1].tolist()
        assert got == want, "read %d" % k


def test_rlo_sortedness_property(hip):
    """RLO build of 2 M reads: strings spelled from increasing rows of rope $ are non-decreasing in
    reverse-lexicographic order (README.md:18-19 identity), and each is a read of the input."""
    n, L = 2_000_000, 101
    dev = hip.HipBwt(1)
    p = dev.dev_alloc(n * (L + 1))
    dev.synth_reads(p, 0, n, L, seed=42)
    dev.insert_multi_dev(p, n * (L + 1))
    dev.dev_free(p)
    c = dev.counts()
    rows = sorted(set([0, 1, 2, n // 3, n // 3 + 1, n // 2, n - 2, n - 1]))
    spelled = [_walk(dev, c, r) for r in rows]      # reversed reads
    assert all(len(s) == L for s in spelled)
    assert all(spelled[i] <= spelled[i + 1] for i in range(len(spelled) - 1))


# ---- sparse layout (leaves with slack, in-place rounds): forced on for every round ----------------------------------

class _ForcedSparse:
    """RB2_SPARSE_LAMBDA is read by rb2_hip_create: a huge threshold puts every round of every batch on the sparse path
    (k_part_sparse / k_merge_leaf / locate()), and RB2_SPARSE_MAXPEN=0 re-enters it right after each dense fallback, so
    re-layouts in both directions and void rounds (leaf overflow) happen all the time."""
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("RB2_SPARSE_LAMBDA", "RB2_SPARSE_MAXPEN")}
        os.environ["RB2_SPARSE_LAMBDA"] = "1e18"
        os.environ["RB2_SPARSE_MAXPEN"] = "0"
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("so", [0, 1, 2])
def test_sparse_forced_repetitive_multibatch(hip, so):
    reads = H.repetitive_reads(1500, seed=11 + so)
    with _ForcedSparse():
        run_both(hip, so, [H.encode_batch(reads[:500]), H.encode_batch(reads[500:1000]), H.encode_batch(reads[1000:], True, True)])


@pytest.mark.parametrize("so", [0, 1, 2])
def test_sparse_forced_random_golden(hip, golden, so):
    g = golden["sets"]["100k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    with _ForcedSparse():
        dev = hip.HipBwt(so)
        for i in range(0, 100000, 40000):
            dev.insert_multi(H.encode_batch_fixed(codes[i:i + 40000]))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_sparse_forced_long_reads(hip, golden, so):
    g = golden["sets"]["200_x_10k"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    with _ForcedSparse():
        dev = hip.HipBwt(so)
        dev.insert_multi(H.encode_batch_fixed(codes))
    assert H.md5(H.bwt_text(dev.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_sparse_forced_two_and_three_plane_leaves(hip, so):
    """long reads with runs of N, three batches, every round in place: leaves without `$` / `N` are two-plane leaves (their third line is neither read
    nor written: rb2_device.h), the ones the N runs and the sentinels of earlier batches fall into keep three -- both kinds next to each other in every
    superblock, leaves that get their first `$` / `N` in the middle of a batch (the flag travels with the fill), leaf splits of both kinds, re-spreads.
    Ropes and counts against the oracle after every batch, then rank queries answered from the in-place layout (the readers of leaf words other
    than the merge: leaf_count / wave_leaf_counts with and without a plane-2 line)"""
    rng = np.random.RandomState(77 + so)
    reads = []
    for i in range(360):
        r = list(rng.randint(1, 5, size=int(rng.randint(1500, 2600))))
        for _ in range(int(rng.randint(0, 3))):                            # 0-2 runs of N per read
            at, n = int(rng.randint(0, len(r) - 1)), int(rng.randint(1, 300))
            r[at:at + n] = [5] * len(r[at:at + n])
        reads.append(r)
    with _ForcedSparse():
        o, dev = H.Oracle(so), hip.HipBwt(so)
        for i, buf in enumerate([H.encode_batch(reads[:150]), H.encode_batch(reads[150:270]), H.encode_batch(reads[270:], True, so == 2)]):
            o.insert_multi(buf); dev.insert_multi(buf)
            assert np.array_equal(o.counts(), dev.counts()), "count matrix differs after batch %d" % i
            st = dev.layout_stats()
            assert st["sparse_now"], st                                     # the queries below are answered from the in-place layout
            for b in range(1, 6):
                r = o.rope(b)
                xs = np.concatenate([[0, 1, len(r) - 1, len(r)], rng.randint(0, len(r) + 1, size=100)])
                got = dev.rank_batch(b, xs)
                cum = np.zeros((len(r) + 1, 6), np.int64)
                for s in range(6):
                    cum[1:, s] = np.cumsum(r == s)
                assert np.array_equal(got, cum[np.minimum(xs, len(r))]), "rank on rope %d after batch %d" % (b, i)
        assert dev.layout_stats()["sparse_rounds"] > 3000
        for b in range(6):                                                  # (the export leaves the in-place layout: last)
            assert np.array_equal(o.rope(b), dev.rope(b)), "rope %d" % b


@pytest.mark.parametrize("so", [0, 1, 2])
def test_sparse_forced_edge_shapes(hip, so):
    lay = hip.HipBwt.layout()
    n = lay["leaf_syms"] * lay["tile_leaves"] + 37
    homo = [[1] * 70] * 300 + [[4] * 33] * 200 + [[2] * 5 + [5] * 3] * 100 + [[3] * (n // 64)] * 70
    with _ForcedSparse():
        run_both(hip, so, [np.zeros(1, np.uint8), np.zeros(7, np.uint8), np.array([3, 0], np.uint8), H.encode_batch([[1], [], [1], [5], [4, 4], []])])
        run_both(hip, so, [H.encode_batch(homo[:400]), H.encode_batch(homo[400:])])
        base = H.splitmix_bases(30000, 101, seed=5)
        small = H.splitmix_bases(700, 101, seed=6)
        dense = [[1] * 90] * 400 + [[2, 2, 2, 3] * 20] * 300
        run_both(hip, so, [H.encode_batch_fixed(base), H.encode_batch_fixed(small), H.encode_batch(dense)])


@pytest.mark.parametrize("so,strand", [(0, 0), (1, 0), (2, 1)])
def test_sparse_forced_coverage_reads(hip, so, strand):
    """overlapping reads: non-empty intervals -> the search-based rank (locate) inside k_prep<false, sparse>"""
    n, L, glen = 12000, 50, 10000
    per = (L + 1) * (2 if strand else 1)
    o = H.Oracle(so)
    with _ForcedSparse():
        dev = hip.HipBwt(so)
    p = dev.dev_alloc(n * per + 64)
    for first in (0, n):
        dev.synth_reads(p, first, n, L, seed=11, strand=strand, genome_len=glen)
        dev.sync()
        host = np.empty(n * per, np.uint8)
        dev.L.rb2_hip_memcpy(dev.h, host.ctypes.data, p, n * per, 1)
        dev.insert_multi_dev(p, n * per)
        o.insert_multi(host)
        assert np.array_equal(dev.counts(), o.counts())
        r = o.rope(1)                                        # rank queries answered on whatever layout the batch ended in
        for x in (0, 1, len(r) // 3, len(r) - 1, len(r)):
            assert np.array_equal(dev.rank1a(1, x), np.bincount(r[:x], minlength=6))
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b)), "rope %d" % b
    dev.dev_free(p)


def test_sparse_forced_fuzz(hip):
    """soak runs: RB2_FUZZ_ITERS=400 RB2_FUZZ_SEED=k"""
    rng = np.random.RandomState(4242 + 7919 * int(os.environ.get("RB2_FUZZ_SEED", "0")))
    with _ForcedSparse():
        for it in range(int(os.environ.get("RB2_FUZZ_ITERS", "12"))):
            so = int(rng.randint(3))
            batches = []
            for _ in range(int(rng.randint(1, 4))):
                n = int(rng.choice([1, 2, 63, 65, 300, 513, 1500, 4000]))
                pool = [list(rng.randint(1, 5, size=rng.randint(1, 30))) for _ in range(5)]
                reads = []
                for _ in range(n):
                    L = int(rng.choice([0, 1, 5, 17, 40, 120]))
                    r = (pool[rng.randint(5)] * 8)[:L] if rng.rand() < 0.3 else list(rng.randint(1, 5, size=L))
                    reads.append([int(x) for x in r])
                batches.append(H.encode_batch(reads, True, rng.rand() < 0.25))
            run_both(hip, so, batches)


def test_sparse_engages_on_long_reads(hip):
    """1000 x 2 kbp in input order with the threshold at lambda < 2: the first ~670 rounds rewrite the (small) index densely,
    then the engine re-lays the index out with slack and inserts in place; equals the oracle, and the statistics say so"""
    codes = H.splitmix_bases(1000, 2000, seed=91)
    old = os.environ.get("RB2_SPARSE_LAMBDA")
    os.environ["RB2_SPARSE_LAMBDA"] = "2.0"
    try:
        o, g = run_both(hip, 0, [H.encode_batch_fixed(codes)])
    finally:
        if old is None:
            os.environ.pop("RB2_SPARSE_LAMBDA", None)
        else:
            os.environ["RB2_SPARSE_LAMBDA"] = old
    st = g.sparse_stats()
    assert st["sparse_rounds"] > 1000 and st["relayouts"] >= 1, st


@pytest.mark.parametrize("sparse", [False, True])
def test_rank_batch_matches_oracle(hip, sparse):
    """rb2_hip_rank_batch (one wave per query, wave_rank_all) on both leaf layouts against a bincount of the oracle's ropes"""
    reads = H.repetitive_reads(6000, seed=8, genome_len=900, max_len=120) + [[1] * 400] * 30
    if sparse:
        with _ForcedSparse():
            o, dev = run_both(hip, 2, [H.encode_batch(reads[:3000]), H.encode_batch(reads[3000:], True, True)])
    else:
        o, dev = run_both(hip, 2, [H.encode_batch(reads[:3000]), H.encode_batch(reads[3000:], True, True)])
    rng = np.random.RandomState(5)
    for b in range(6):
        r = o.rope(b)
        xs = np.concatenate([[0, 1, len(r) - 1, len(r), len(r) + 5], rng.randint(0, len(r) + 1, size=300)]).clip(0, None)
        got = dev.rank_batch(b, xs)
        cum = np.zeros((len(r) + 1, 6), np.int64)
        for s in range(6):
            cum[1:, s] = np.cumsum(r == s)
        assert np.array_equal(got, cum[np.minimum(xs, len(r))]), "rope %d" % b
        assert np.array_equal(dev.rank1a(b, int(xs[7])), got[7])


@pytest.mark.parametrize("so", [1, 2])
def test_huge_groups_long_intervals(hip, so):
    """amplicon-like input: thousands of identical and near-identical reads -> every round has intervals of thousands of rows,
    most of them spanning leaves: the wave-cooperative interval counts of k_prep (wave_range_counts) do the work"""
    rng = np.random.RandomState(12)
    base = list(rng.randint(1, 5, size=150))
    reads = []
    for i in range(9000):
        r = list(base)
        if i % 3 == 0:
            r[int(rng.randint(150))] = int(rng.randint(1, 5))
        reads.append(r[int(rng.randint(0, 20)):])
    run_both(hip, so, [H.encode_batch(reads[:4000]), H.encode_batch(reads[4000:], True, so == 2)])


@pytest.mark.parametrize("seed,lam", [(11, "3.0"), (12, "8.0"), (13, "5.0")])
def test_fuzz_medium_jobs_with_layout_changes(hip, seed, lam):
    """medium-size jobs (0.1-0.25 M reads in 3-4 batches, i.i.d. or overlapping, random order / strands) with the sparse threshold
    raised so that the layout changes by itself a few times per job: thousands of leaves are re-laid out with slack, tens of
    thousands of in-place rounds' worth of leaves are touched, void rounds send the index back to the dense layout"""
    rng = np.random.RandomState(900 + seed)
    so, strand = int(rng.randint(3)), int(rng.randint(2))
    L = int(rng.randint(60, 200))
    nb = int(rng.randint(3, 5))
    n = int(rng.randint(100000, 250001)) // nb
    cov = float(rng.choice([0, 10, 40]))
    glen = int(max(L + 1, n * nb * L / cov)) if cov else 0
    per = (L + 1) * (2 if strand else 1)
    old = os.environ.get("RB2_SPARSE_LAMBDA")
    os.environ["RB2_SPARSE_LAMBDA"] = lam
    try:
        o, dev = H.Oracle(so), hip.HipBwt(so)
    finally:
        if old is None:
            os.environ.pop("RB2_SPARSE_LAMBDA", None)
        else:
            os.environ["RB2_SPARSE_LAMBDA"] = old
    p = dev.dev_alloc(n * per + 64)
    for b in range(nb):
        dev.synth_reads(p, b * n, n, L, seed=300 + seed, strand=strand, genome_len=glen)
        dev.sync()
        host = np.empty(n * per, np.uint8)
        dev.L.rb2_hip_memcpy(dev.h, host.ctypes.data, p, n * per, 1)
        dev.insert_multi_dev(p, n * per)
        o.insert_multi(host)
        assert np.array_equal(dev.counts(), o.counts()), "counts after batch %d (so %d strand %d L %d cov %g)" % (b, so, strand, L, cov)
    st = dev.sparse_stats()
    for r in range(6):
        assert np.array_equal(dev.rope(r), o.rope(r)), "rope %d (so %d strand %d L %d cov %g, %s)" % (r, so, strand, L, cov, st)
    assert st["sparse_rounds"] > 0, st
    dev.dev_free(p)


def test_lazy_host_insert_overlaps_and_stays_exact(hip):
    """rb2_hip_insert_multi may return while its rounds run (rb2_hip_set_lazy, the default): batches fired back to back -- the next
    text is uploaded to the second device buffer beside the kernels --, queries in between (every entry point waits first), the count
    matrix of a batch from its text alone (rb2_hip_last_batch_counts = what mr_insert_multi adds to mr_get_c), all against the oracle"""
    import helpers as H
    for so in (0, 1, 2):
        o = H.Oracle(so)
        dev = hip.HipBwt(so)
        sync = hip.HipBwt(so)
        sync.set_lazy(0)
        bufs = [H.encode_batch_fixed(H.splitmix_bases(3000, 80, seed=11)), H.encode_batch(H.repetitive_reads(1500, seed=5, genome_len=700, max_len=90), True, so == 2),
                H.encode_batch([[1, 2, 3], [], [4] * 50, [5, 5]]), H.encode_batch_fixed(H.splitmix_bases(2000, 101, seed=12))]
        for i, buf in enumerate(bufs):
            before = o.counts().copy()
            o.insert_multi(buf)
            keep = buf.copy()
            dev.insert_multi(buf)
            d = dev.last_batch_counts()
            assert d is not None and np.array_equal(d, o.counts() - before), i        # valid at once, no wait
            buf[:] = 0                                                                 # the buffer is the caller's again
            sync.insert_multi(keep)
            if i == 1:                                                                 # a query in between sees the finished batch
                assert np.array_equal(dev.counts(), o.counts())
                assert np.array_equal(dev.rank1a(2, 17), np.bincount(o.rope(2)[:17], minlength=6))
        assert np.array_equal(dev.counts(), o.counts()) and np.array_equal(sync.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(dev.rope(b), o.rope(b)) and np.array_equal(sync.rope(b), o.rope(b)), (so, b)
        # a device-buffer insert invalidates the per-batch matrix
        p = dev.dev_alloc(64)
        z = np.zeros(3, np.uint8); z[:2] = (1, 2)
        dev.L.rb2_hip_memcpy(dev.h, p, z.ctypes.data, 3, 0)
        dev.insert_multi_dev(p, 3)
        assert dev.last_batch_counts() is None
        dev.dev_free(p); dev.close(); sync.close()


@pytest.mark.parametrize("so", [0, 1, 2])
def test_batch_with_too_many_strings_is_cut_not_refused(hip, so):
    """a batch may hold 2^32 - 1024 strings; the reference takes any count (mrope.c:269-277).  More: the engine cuts the batch at a
    sentinel near its middle and inserts the halves one after the other (recursively) -- the same BWT.  Limit lowered for the test."""
    codes = H.splitmix_bases(5000, 37, seed=77)
    reads = H.repetitive_reads(1500, seed=9, genome_len=400, max_len=60)
    bufs = [H.encode_batch_fixed(codes), H.encode_batch(reads, True, True)]
    o = H.Oracle(so)
    for b in bufs:
        o.insert_multi(b)
    os.environ["RB2_MAX_BATCH_STRINGS"] = "700"
    try:
        g = hip.HipBwt(so, 0)
        for b in bufs:
            g.insert_multi(b)
    finally:
        del os.environ["RB2_MAX_BATCH_STRINGS"]
    assert np.array_equal(o.counts(), g.counts())
    for b in range(6):
        assert np.array_equal(o.rope(b), g.rope(b)), "rope %d" % b
    g.close()


@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("mode", ["widen0", "widen3", "widen17", "wide", "widen5_sparse"])
def test_position_storage_width(hip, so, mode):
    """The per-string positions are stored as 32-bit values while no sub-rope can hold 2^32 symbols and widened the round before
    one could (maybe_widen).  Forced here: leave the narrow mode before round 0 / 3 / 17 of every batch, never enter it
    (RB2_POS=64), and widen in the middle of in-place rounds -- ropes and count matrix against the oracle."""
    env = {"widen0": {"RB2_POS_WIDEN_AT": "0"}, "widen3": {"RB2_POS_WIDEN_AT": "3"}, "widen17": {"RB2_POS_WIDEN_AT": "17"}, "wide": {"RB2_POS": "64"},
           "widen5_sparse": {"RB2_POS_WIDEN_AT": "5", "RB2_SPARSE_LAMBDA": "1e18", "RB2_SPARSE_MAXPEN": "0", "RB2_SPARSE_HEAD": "2"}}[mode]
    reads = H.repetitive_reads(2500, seed=31 + so, genome_len=700, max_len=90)
    codes = H.splitmix_bases(3000, 60, seed=5)
    bufs = [H.encode_batch(reads[:1500]), H.encode_batch_fixed(codes), H.encode_batch(reads[1500:], True, True)]
    o = H.Oracle(so)
    os.environ.update(env)
    try:
        g = hip.HipBwt(so, 0)
        for b in bufs:
            o.insert_multi(b); g.insert_multi(b)
            assert np.array_equal(o.counts(), g.counts())
    finally:
        for k in env:
            del os.environ[k]
    for b in range(6):
        assert np.array_equal(o.rope(b), g.rope(b)), "rope %d" % b
    g.close()
