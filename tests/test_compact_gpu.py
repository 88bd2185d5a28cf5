"""Compact windows of the dense layout (csrc/rb2_merge.h "window formats"): two bit planes + the positions of the `$` / `N`
symbols instead of a third plane, chosen per window of 4096 symbols by the merge itself, plain (three planes) again wherever
anything but the merge reads the pool.  Everything bit-exact against the oracle, and the per-format window counts
(RB2_COMPACT_STATS=1) show that the formats the test is about were really written and read back."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


class Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(dev, o):
    assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        ro, rg = o.rope(b), dev.rope(b)
        assert len(ro) == len(rg), "rope %d length" % b
        assert np.array_equal(ro, rg), "rope %d differs at %s" % (b, np.flatnonzero(ro != rg)[:5])


def _run(hip, so, batches, **env):
    with Env(RB2_COMPACT_STATS=1, RB2_SPARSE_LAMBDA=0, **env):      # (stay dense: this is about the dense merge)
        dev, o = hip.HipBwt(so), H.Oracle(so)
        for buf in batches:
            o.insert_multi(buf)
            dev.insert_multi(buf)
            _same(dev, o)                                           # after EVERY batch: its last round must have left plain windows
        st = dev.window_stats()
        dev.close()
    return st


@pytest.mark.parametrize("so", [0, 1, 2])
def test_all_formats_in_one_index(hip, so):
    """random reads (1 % sentinels: one exception line per window), a batch of very short reads and runs of N (windows with more
    than 127 exceptions stay plain, next to compact ones of the same piece), 40 copies of one read, then random reads again on top
    of that mixture; the first batch on the empty index has no exception at all until its last round"""
    base = H.splitmix_bases(60000, 101, seed=21)
    rng = np.random.default_rng(5 + so)
    short = [list(rng.integers(1, 5, size=int(n))) for n in rng.integers(1, 7, size=30000)]
    nruns = [[5] * int(n) + list(rng.integers(1, 5, size=40)) for n in rng.integers(150, 400, size=300)]
    dup = [list(rng.integers(1, 5, size=101))] * 40
    long_mixed = [list(rng.integers(1, 5, size=101)) for _ in range(4000)]
    mixed = short + nruns + dup + long_mixed
    order = rng.permutation(len(mixed))
    mixed = [mixed[i] for i in order]
    more = H.splitmix_bases(50000, 101, seed=22)
    st = _run(hip, so, [H.encode_batch_fixed(base[:30000]), H.encode_batch_fixed(base[30000:]), H.encode_batch(mixed), H.encode_batch_fixed(more, True, so == 2)])
    assert st["counted"] and st["compact_rounds"] > 150, st
    assert st["compact0"] > 0 and st["compact1"] > 0 and st["plain"] > 0, st


def test_two_exception_lines_and_overflow_to_plain(hip):
    """reads of 40 bp: one sentinel per 41 symbols, about 100 exceptions per window -- most windows need the second line,
    some pass 127 and stay plain"""
    r40 = H.splitmix_bases(40000, 40, seed=31)
    r101 = H.splitmix_bases(20000, 101, seed=32)
    st = _run(hip, 0, [H.encode_batch_fixed(r40[:20000]), H.encode_batch_fixed(r40[20000:]), H.encode_batch_fixed(r101)])
    assert st["compact2"] > 0 and st["compact1"] > 0 and st["plain"] > 0, st


@pytest.mark.parametrize("so", [0, 1])
def test_compact_off_is_the_same_index(hip, so):
    base = H.splitmix_bases(30000, 101, seed=41)
    bufs = [H.encode_batch_fixed(base[:10000]), H.encode_batch_fixed(base[10000:])]
    st = _run(hip, so, bufs, RB2_COMPACT=0)
    assert st["compact_rounds"] == 0 and st["compact0"] + st["compact1"] + st["compact2"] == 0, st


def _handover(hip, make, so, lam):
    """two batches of long reads with the switch to the in-place layout in the MIDDLE of each: RB2_SPARSE_LAMBDA = lam makes a batch of m reads
    run dense (compact windows: every interval is empty) until the index holds about 1024 m / lam symbols per read set, then one round
    that writes PLAIN windows over the compact ones (the re-layout reads plain leaves), then the re-layout and in-place rounds"""
    a = H.splitmix_bases(300, 3000, seed=51)
    b = H.splitmix_bases(300, 2500, seed=52)
    with Env(RB2_COMPACT_STATS=1, RB2_SPARSE_LAMBDA=lam):
        dev, o = make(so), H.Oracle(so)
        for buf in (H.encode_batch_fixed(a), H.encode_batch_fixed(b)):
            o.insert_multi(buf); dev.insert_multi(buf)
            _same(dev, o)                                          # after every batch
        st, ls = dev.window_stats(), dev.layout_stats()
        dev.close()
    return st, ls


@pytest.mark.parametrize("so", [0, 1])
def test_leaving_the_dense_layout_from_compact_windows(hip, so):
    st, ls = _handover(hip, hip.HipBwt, so, "20")
    # ~50 dense rounds at the head of the first batch, 8 at the head of the second (the index is in the in-place layout when it starts): compact windows
    # were written and read back (compact0: the first batch has no `$` until its last round), and a switch went through a round of plain windows
    assert st["compact_rounds"] >= 20 and st["compact0"] + st["compact1"] > 0, st
    assert ls["plain_handovers"] >= 1, ls
    assert ls["sparse_rounds"] > 4000 and ls["relayouts"] >= 3, ls


@pytest.mark.parametrize("n", [2, 8])
def test_leaving_the_dense_layout_from_compact_windows_on_virtual_ranks(hip, n):
    """the same hand-over on every rank of a sharded index (ranks decide on their own layout: their share of the strings against their leaves)"""
    from ropebwt2_amd.hipbwt import MultiBwt
    a = H.splitmix_bases(300, 3000, seed=51)
    b = H.splitmix_bases(300, 2500, seed=52)
    with Env(RB2_COMPACT_STATS=1, RB2_SPARSE_LAMBDA="20"):
        m, o = MultiBwt(0, [0] * n, "peer"), H.Oracle(0)
        for buf in (H.encode_batch_fixed(a), H.encode_batch_fixed(b)):
            o.insert_multi(buf); m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts())
            for r in range(6):
                assert np.array_equal(m.rope(r), o.rope(r)), "rope %d" % r
        ws = [m.engine(k).window_stats() for k in range(n)]
        ls = [m.engine(k).layout_stats() for k in range(n)]
        m.close()
    assert sum(w["compact_rounds"] for w in ws) >= 20 and sum(w["compact0"] + w["compact1"] for w in ws) > 0, ws
    assert sum(l["plain_handovers"] for l in ls) >= 1 and sum(l["sparse_rounds"] for l in ls) > 4000, ls


def test_compact_windows_are_never_reached_when_the_layout_switches_at_once(hip):
    """(the round-5 form of the test above: with RB2_SPARSE_LAMBDA = 1e18 the index leaves the dense layout in its first round, before a compact window exists)"""
    st, ls = _handover(hip, hip.HipBwt, 0, "1e18")
    assert ls["sparse_rounds"] > 4000 and ls["plain_handovers"] == 0, (st, ls)


def test_configs1_shape_2M_reads_compact(hip):
    """2 M x 101 bp in two batches through the device-side generator: LF-walks spell the reads (as test_hip_parity does at this
    size), with compact windows in every all-empty round"""
    n, L = 2_000_000, 101
    with Env(RB2_COMPACT_STATS=1):
        dev = hip.HipBwt(1)
        for first, cnt in ((0, 1_200_000), (1_200_000, 800_000)):
            p = dev.dev_alloc(cnt * (L + 1))
            dev.synth_reads(p, first, cnt, L, seed=42)
            dev.insert_multi_dev(p, cnt * (L + 1))
            dev.dev_free(p)
        st = dev.window_stats()
        c = dev.counts()
        assert int(c.sum()) == n * (L + 1)
        hs = dev.rope_hashes()
    with Env(RB2_COMPACT=0):
        ref = hip.HipBwt(1)
        for first, cnt in ((0, 1_200_000), (1_200_000, 800_000)):
            p = ref.dev_alloc(cnt * (L + 1))
            ref.synth_reads(p, first, cnt, L, seed=42)
            ref.insert_multi_dev(p, cnt * (L + 1))
            ref.dev_free(p)
        assert np.array_equal(ref.counts(), c)
        assert ref.rope_hashes() == hs
        ref.close()
    dev.close()
    assert st["compact0"] > 0 and st["compact1"] > 0, st
