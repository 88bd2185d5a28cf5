"""In-place rounds that do not fall off a cliff: the leaf split of the sparse layout (k_split -- split_node / rope.c:78-112,
143-146 re-derived for slack leaves), re-spreads instead of dense detours when a superblock runs out of slots, and in-place
rounds on a SHARDED index (rb2_hip_multi_*).  Everything bit-exact against the oracle."""
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


FORCED = dict(RB2_SPARSE_LAMBDA="1e18", RB2_SPARSE_MAXPEN="0")


def _same(dev, o):
    assert np.array_equal(dev.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(dev.rope(b), o.rope(b)), "rope %d" % b


@pytest.mark.parametrize("so", [0, 1, 2])
def test_index_grows_by_splitting_leaves(hip, so):
    """an index that grows 30-fold while it stays in the sparse layout: leaves split into their superblock's reserve slots,
    full superblocks trigger a re-spread; no void round after the hot-spot head of each batch"""
    first = H.splitmix_bases(150, 400, seed=4)
    rest = H.splitmix_bases(1200, 1500, seed=5)
    with Env(**FORCED):
        dev = hip.HipBwt(so)
        o = H.Oracle(so)
        for buf in (H.encode_batch_fixed(first), H.encode_batch_fixed(rest[:600]), H.encode_batch_fixed(rest[600:], True, so == 2)):
            o.insert_multi(buf)
            dev.insert_multi(buf)
        st = dev.layout_stats()
        for b in range(1, 5):                                        # rank queries on the split layout
            ro = o.rope(b)
            xs = np.linspace(0, len(ro), 9).astype(np.int64)
            want = np.array([np.bincount(ro[:x], minlength=6) for x in xs])
            assert np.array_equal(dev.rank_batch(b, xs), want)
        _same(dev, o)
        dev.close()
    assert st["leaf_splits"] > 500, st
    assert st["sparse_rounds"] > 0.9 * (401 + 2 * 1501), st


def test_homopolymer_stays_in_place(hip):
    """one string of 60,000 A into a sparse index: every round inserts one symbol next to the last one -- the same leaf over and over.
    Before: a void round every 336 symbols and a dense detour each time; now the leaf splits, then its superblock is re-spread."""
    base = H.splitmix_bases(3000, 200, seed=8)
    n_a = 60_000
    with Env(RB2_SPARSE_LAMBDA="1e18"):                              # (default back-off policy: a detour would show)
        for so in (0, 1):
            dev, o = hip.HipBwt(so), H.Oracle(so)
            b0 = H.encode_batch_fixed(base)
            o.insert_multi(b0); dev.insert_multi(b0)
            s0 = dev.layout_stats()
            hp = H.encode_batch([[1] * n_a])
            o.insert_multi(hp); dev.insert_multi(hp)
            st = dev.layout_stats()
            _same(dev, o)
            dev.close()
            rounds = n_a + 1
            assert st["void_rounds"] == s0["void_rounds"], (s0, st)
            assert st["sparse_rounds"] - s0["sparse_rounds"] >= rounds - 16, (s0, st)
            assert st["leaf_splits"] - s0["leaf_splits"] >= n_a // 1344, (s0, st)


def test_many_strings_into_one_leaf_is_a_void_round_and_still_right(hip):
    """more inserts into one leaf than it can ever take (thousands of identical strings): the round is void, redone densely"""
    base = H.splitmix_bases(2000, 120, seed=9)
    dup = [[1, 2, 3, 4] * 10] * 5000
    with Env(**FORCED):
        for so in (0, 1, 2):
            dev, o = hip.HipBwt(so), H.Oracle(so)
            for buf in (H.encode_batch_fixed(base), H.encode_batch(dup)):
                o.insert_multi(buf); dev.insert_multi(buf)
            _same(dev, o)
            assert dev.layout_stats()["void_rounds"] > 0
            dev.close()


def test_void_round_behind_a_many_tiles_look_ahead(hip):
    """the counting phase of round r + 1 is queued before the host sees the verdict of in-place round r; with many string tiles its
    k_setup rides on k_tfix (not on the single-block k_tscan_setup) and must not replace the descriptors of a round that turns out
    void (RB2_TS_MAX lowers the tile count at which the many-tiles kernels take over: 5000 strings are 10 tiles)"""
    base = H.splitmix_bases(2000, 120, seed=9)
    dup = [[1, 2, 3, 4] * 10] * 5000
    more = H.splitmix_bases(3000, 150, seed=10)
    with Env(RB2_TS_MAX=2, **FORCED):
        for so in (0, 1, 2):
            dev, o = hip.HipBwt(so), H.Oracle(so)
            for buf in (H.encode_batch_fixed(base), H.encode_batch(dup), H.encode_batch_fixed(more)):
                o.insert_multi(buf); dev.insert_multi(buf)
            _same(dev, o)
            assert dev.layout_stats()["void_rounds"] > 0
            dev.close()


@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("n", [2, 8])
def test_sharded_index_inserts_in_place(hip, so, n):
    """in-place rounds behind the one-handle API: every rank picks the layout of its own slice; forced on here"""
    from ropebwt2_amd import MultiBwt
    reads = H.repetitive_reads(2500, seed=40 + so, genome_len=900, max_len=120)
    codes = H.splitmix_bases(900, 700, seed=6)
    with Env(**FORCED):
        o = H.Oracle(so)
        m = MultiBwt(so, [0] * n, "peer")
        for buf in (H.encode_batch(reads[:1500]), H.encode_batch_fixed(codes), H.encode_batch(reads[1500:], True, True)):
            o.insert_multi(buf)
            m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts())
        st = m.stats()
        for b in range(6):
            assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
        m.close()
    assert st["sparse_rounds"] > 0.5 * n * st["rounds"] / 2, st


def test_sharded_configs4_shape_runs_sparse(hip):
    """configs[4]'s regime on 8 virtual ranks: few new strings onto a loaded index (lambda far below the threshold) -- the default
    policy puts the rounds in place on every rank that holds a heavy piece; ropes equal the oracle's"""
    from ropebwt2_amd import MultiBwt
    from ropebwt2_amd.hipbwt import encode_runs
    old = H.splitmix_bases(60000, 100, seed=3)
    new = H.splitmix_bases(300, 100, seed=4)
    o = H.Oracle(1)
    o.insert_multi(H.encode_batch_fixed(old))
    m = MultiBwt(1, [0] * 8, "peer")
    m.load_ropes([encode_runs(o.rope(b)) for b in range(6)])
    for i in range(0, 300, 100):
        buf = H.encode_batch_fixed(new[i:i + 100])
        o.insert_multi(buf)
        m.insert_multi(buf)
    st = m.stats()
    assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    m.close()
    assert st["sparse_rounds"] > 8 * 3 * 60, st
