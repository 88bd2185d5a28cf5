"""BASELINE.json's configs[2], [3] and [4] AT FULL SIZE on one MI355X, inside `pytest -m gpu` (about three minutes together).
Inputs are generated on the device (the splitmix64 stream of SURVEY.md 8c) just before they are inserted.

  configs[2]  1.2 B x 101 bp, -brR (RCLO, forward strand), -m10g: 12 batches, 122.4 G symbols on ONE engine; the index after
              the first three batches is compared -- rope by rope, through device-side checksums of the packed symbols
              (rb2_hip_rope_hash) and the count matrix -- with the same three batches built by 8 virtual ranks behind one handle
              (rb2_hip_multi_*, the owner map an 8-GPU run uses); size-independent properties at the end.
              The ORDER is pinned to the real reference by the 100 M-read RCLO golden (golden_large.json configs2_order_100M)
              through the CLI on 8 ranks.
  configs[3]  10 M x 10 kbp, input order: 10 batches of 10,001 rounds, 100 G symbols; LF consistency, rope $ = last bases of
              the reads, inverse-BWT walks reproduce sampled reads, layout statistics (rounds in place).
  configs[4]  500 M-read index exported to run bytes -> rb2_hip_load_ropes into a fresh handle (decoded on the device) -> 500 M
              more reads == the one-shot build of all 10^9 reads (rope checksums + count matrix).

Integer/byte work: every comparison is exact."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def batch_reads(mem_gib, read_len):
    m = int(mem_gib * 1024 ** 3 * 0.97) + 1                  # main.c:136
    return -(-m // (read_len + 1))                           # main.c:238


def _lf_ok(c, n_strings):
    sizes, occ = c.sum(axis=1), c.sum(axis=0)
    return int(sizes[0]) == n_strings and int(occ[0]) == n_strings and all(int(sizes[a]) == int(occ[a]) for a in range(1, 6))


def _sym_at(dev, b, p):
    return int(np.argmax(dev.rank1a(b, p + 1) - dev.rank1a(b, p)))


def _walk(dev, c, row, limit):
    """inverse-BWT walk from row `row` of rope $: the symbols of one string, last base first, until its sentinel"""
    b, got = 0, []
    while len(got) <= limit:
        s = _sym_at(dev, b, row)
        if s == 0:
            return got
        got.append(s)
        row = int(c[:b, s].sum() + dev.rank1a(b, row)[s])
        b = s
    return None


def test_rope_hash_is_a_checksum_of_the_symbols(hip):
    """rb2_hip_rope_hash: equal for one engine, N ranks behind one handle and an index loaded from run bytes; differs when
    two symbols swap places"""
    from ropebwt2_amd import MultiBwt
    from ropebwt2_amd.hipbwt import encode_runs
    reads = H.repetitive_reads(3000, seed=5, genome_len=900, max_len=120)
    bufs = [H.encode_batch(reads[:2000]), H.encode_batch(reads[2000:], True, True)]
    for so in (0, 2):
        one, m = hip.HipBwt(so), MultiBwt(so, [0] * 8, "peer")
        o = H.Oracle(so)
        for b in bufs:
            one.insert_multi(b); m.insert_multi(b); o.insert_multi(b)
        h1 = one.rope_hashes()
        assert h1 == m.rope_hashes()
        ld = hip.HipBwt(so)
        ropes = [o.rope(b) for b in range(6)]
        ld.load_ropes([encode_runs(r) for r in ropes])
        assert ld.rope_hashes() == h1
        r = ropes[1].copy()
        i = int(np.flatnonzero(r[:-1] != r[1:])[0])
        r[i], r[i + 1] = r[i + 1], r[i]                                  # same counts, other order
        ld.load_ropes([encode_runs(x) for x in [ropes[0], r] + ropes[2:]])
        h2 = ld.rope_hashes()
        assert h2[1] != h1[1] and h2[0] == h1[0] and h2[2:] == h1[2:]
        for x in (one, m, ld):
            x.close()


def test_configs2_full_size_1200M_rclo(hip):
    from ropebwt2_amd import MultiBwt
    L, N = 101, 1_200_000_000
    per = batch_reads(10, L)
    so = 2
    # (a) the first three batches on 8 virtual ranks behind one handle
    m = MultiBwt(so, [0] * 8, "peer")
    e0 = m.engine(0)
    buf = e0.dev_alloc(per * (L + 1) + 64)
    marks = []
    for k in range(3):
        e0.synth_reads(buf, k * per, per, L, seed=42)
        e0.sync()
        m.insert_multi_dev(buf, per * (L + 1))
        marks.append((m.counts().copy(), None))
    marks[-1] = (marks[-1][0], m.rope_hashes())
    st = m.stats()
    assert st["host_syncs_in_rounds"] == 0 and st["rounds"] == 3 * (L + 1)
    e0.dev_free(buf)
    m.close()
    # (b) the whole job on one engine
    dev = hip.HipBwt(so)
    dev.reserve(per * (L + 1), per, N * (L + 1))
    buf = dev.dev_alloc(per * (L + 1) + 64)
    done, k = 0, 0
    while done < N:
        n = min(per, N - done)
        dev.synth_reads(buf, done, n, L, seed=42)
        dev.sync()
        dev.insert_multi_dev(buf, n * (L + 1))
        done += n
        if k < 3:
            assert np.array_equal(dev.counts(), marks[k][0]), "count matrix after batch %d: 8 ranks vs one engine" % k
            if marks[k][1] is not None:
                assert dev.rope_hashes() == marks[k][1], "ropes after batch %d: 8 ranks vs one engine" % k
        k += 1
    dev.dev_free(buf)
    assert k == 12
    c = dev.counts()
    assert int(c.sum()) == N * (L + 1) and _lf_ok(c, N)
    sizes = c.sum(axis=1)
    rng = np.random.RandomState(2)
    for b in range(1, 5):                                                   # rank: end of rope == matrix row, monotone, sums to the position
        assert np.array_equal(dev.rank1a(b, int(sizes[b])), c[b])
        prev = np.zeros(6, np.int64)
        for x in np.sort(rng.randint(0, int(sizes[b]) + 1, size=8)):
            r = dev.rank1a(b, int(x))
            assert int(r.sum()) == int(x) and np.all(r >= prev)
            prev = r
    for row in (0, N // 3, N - 1):                                          # every row of rope $ starts a walk of exactly L bases
        w = _walk(dev, c, row, L)
        assert w is not None and len(w) == L
    dev.close()


def test_configs2_order_golden_100M_through_cli_on_8_ranks(hip):
    """RCLO at scale against the REAL reference: 100 M x 101 bp, `-LRdr` (configs[2]'s order and strand flags, ropebwt2's default
    -m10g), built by the CLI with the index sharded over 8 ranks behind mr_insert_multi; md5 of the 6 GB .fmd"""
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json"))).get("configs2_order_100M")
    if g is None:
        pytest.skip("golden_large.json has no configs2_order_100M entry (make_golden_large.py --configs2-order)")
    from test_host_layer import CLI
    pg = subprocess.Popen([H.GEN, str(g["n_reads"]), str(g["read_len"]), str(g["seed"])], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI] + g["flags"].split() + ["-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          env=dict(os.environ, RB2_HIP_DEVICES="0,0,0,0,0,0,0,0"))
    h, n = hashlib.md5(), 0
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk); n += len(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert n == g["fmd_bytes"] and h.hexdigest() == g["fmd_md5"]


@pytest.mark.parametrize("flags", ["-LRds", "-LRdr"])
def test_skewed_composition_crosses_2_to_32_symbols_per_subrope(hip, flags):
    """The > 2^32 regime pinned to the REAL reference: 60 M x 101 bp of skewed composition (85 % A), so that sub-rope (A,A) -- the
    A's of rope A, one piece of the device index -- holds 4.35 G symbols: piece-relative positions, window offsets (k_part /
    merge_window: e - i0), the 48-bit fields of the exchange records and the directory prefixes all pass 32 bits.  The .fmd md5 was
    produced by oracle/_ref/ropebwt2 (make_golden_large.py --skewed).  Through the CLI on one engine AND sharded over 8 ranks; the
    size of the piece is asserted on an index built from the same stream generated on the device."""
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json"))).get("skewed_60M")
    if g is None or flags not in g["runs"]:
        pytest.skip("golden_large.json has no skewed_60M entry for %s (make_golden_large.py --skewed)" % flags)
    from test_host_layer import CLI
    N, L = g["n_reads"], g["read_len"]
    for devs in (None, "0,0,0,0,0,0,0,0"):
        env = dict(os.environ)
        if devs:
            env["RB2_HIP_DEVICES"] = devs
        pg = subprocess.Popen([H.GEN, str(N), str(L), str(g["seed"]), "0", "0", "0", str(g["skew"])], stdout=subprocess.PIPE)
        pc = subprocess.Popen([CLI] + flags.split() + ["-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        h, n = hashlib.md5(), 0
        for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
            h.update(chunk); n += len(chunk)
        assert pc.wait() == 0 and pg.wait() == 0
        assert n == g["runs"][flags]["fmd_bytes"] and h.hexdigest() == g["runs"][flags]["fmd_md5"], "devices: %s" % devs
    if flags == "-LRds":                                                    # the piece really is that large (same stream, generated on the device)
        dev = hip.HipBwt(1)
        buf = dev.dev_alloc(N * (L + 1) + 64)
        dev.synth_reads(buf, 0, N, L, seed=g["seed"], skew=g["skew"])
        dev.sync()
        dev.insert_multi_dev(buf, N * (L + 1))
        c = dev.counts()
        dev.dev_free(buf); dev.close()
        assert int(c[1, 1]) > (1 << 32), "piece (A,A) = the A's of rope A: %d symbols" % int(c[1, 1])
        assert int(c.sum()) == N * (L + 1) and _lf_ok(c, N)


def test_configs3_full_size_10M_x_10k(hip):
    n, L, per = 10_000_000, 10_000, batch_reads(10, 10_000)
    dev = hip.HipBwt(0)
    buf = dev.dev_alloc(per * (L + 1) + 64)
    done = 0
    while done < n:
        k = min(per, n - done)
        dev.synth_reads(buf, done, k, L, seed=44)
        dev.sync()
        dev.insert_multi_dev(buf, k * (L + 1))
        done += k
    dev.dev_free(buf)
    c = dev.counts()
    assert int(c.sum()) == n * (L + 1) and _lf_ok(c, n)
    st = dev.sparse_stats()
    rounds = 10 * (L + 1)
    assert st["sparse_rounds"] > 0.9 * rounds, st                           # the long-string path: rounds insert in place
    picks = [0, 1, 2, 4_999_999, n - 1]
    last = np.array([H.splitmix_bases(1, L, seed=44, first=k)[0][-1] for k in picks])
    rope0 = dev.rope(0)
    assert len(rope0) == n and np.array_equal(rope0[picks], last)           # input order: row k of rope $ belongs to read k
    for k in (0, 7_777_777):
        want = H.splitmix_bases(1, L, seed=44, first=k)[0][::-1]
        got = _walk(dev, c, k, L)
        assert got is not None and np.array_equal(np.array(got, np.uint8), want), "read %d" % k
    dev.close()


def test_configs4_full_size_500M_plus_500M_device_api(hip):
    L, half = 101, 500_000_000
    per = batch_reads(10, L)

    def feed(dev, first, count):
        buf = dev.dev_alloc(per * (L + 1) + 64)
        done = 0
        while done < count:
            k = min(per, count - done)
            dev.synth_reads(buf, first + done, k, L, seed=42 if first == 0 else 43)
            dev.sync()
            dev.insert_multi_dev(buf, k * (L + 1))
            done += k
        dev.dev_free(buf)

    a = hip.HipBwt(1)
    a.reserve(per * (L + 1), per, 2 * half * (L + 1))
    feed(a, 0, half)
    c_half, h_half = a.counts().copy(), a.rope_hashes()
    rles = [a.rope_rle(b) for b in range(6)]                                  # the existing index leaves the device as run bytes (what an .fmr holds)
    feed(a, half, half)                                                       # one handle sees all 10^9 reads
    c_all, h_all = a.counts().copy(), a.rope_hashes()
    a.close()
    assert int(c_all.sum()) == 2 * half * (L + 1) and _lf_ok(c_all, 2 * half)
    b = hip.HipBwt(1)
    b.load_ropes(rles)                                                        # decoded into packed leaves on the device (k_ld_*)
    del rles
    assert np.array_equal(b.counts(), c_half) and b.rope_hashes() == h_half
    feed(b, half, half)
    assert np.array_equal(b.counts(), c_all)
    assert b.rope_hashes() == h_all, "incremental build on a loaded index differs from the build that saw all reads"
    b.close()
