"""Rope-sharded multi-GPU path (ropebwt2_amd/sharded.py, rb2_hip_shard_*).

CPU  : the exchange plan and the torch.distributed driver with world_size 2 and 3 over gloo (toy engine).
GPU  : N virtual ranks on one device, bit-exact vs the oracle; two real processes over gloo sharing the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))


def launch(n, args, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "sharded_worker.py")] + [str(a) for a in args]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)


def test_subrope_indexing():
    from ropebwt2_amd.sharded import NR, rope_sym, rope_prev, rope_of
    assert NR == 31 and rope_sym(0) == 0 and rope_prev(0) == 0
    seen = set()
    for b in range(1, 6):
        for x in range(6):
            r = rope_of(b, x)
            assert 1 <= r < NR and rope_sym(r) == b and rope_prev(r) == x
            seen.add(r)
    assert len(seen) == 30
    # pieces of one rope are contiguous and ordered by x: concatenating pieces in index order gives rope order
    assert [rope_sym(r) for r in range(NR)] == sorted(rope_sym(r) for r in range(NR))


def test_exchange_layout_is_consistent():
    from ropebwt2_amd.sharded import NR, default_owners, exchange_layout, rope_sym, rope_prev
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 4, 8, 16, 20):
        owner = default_owners(n)
        assert len(owner) == NR and max(owner) < n
        if n <= 16:
            assert len(set(owner)) == n                    # every rank up to 16 carries load
        g = rng.randint(0, 1000, size=(NR, 6))
        sent = np.array([exchange_layout(owner, n, s, g) for s in range(n)])
        # everything that inserts a symbol 1..5 is sent exactly once, to the owner of piece (a, b)
        assert sent.sum() == g[:, 1:].sum()
        for d in range(n):
            want = 0
            for r in range(NR):
                for a in range(1, 6):
                    r2 = 1 + (a - 1) * 6 + rope_sym(r)
                    if owner[r2] == d:
                        want += g[r, a]
            assert sent[:, d].sum() == want


def test_default_owners_balance_dna():
    """uniform DNA: 16 heavy pieces; the heaviest rank holds at most ceil(16/n) of them"""
    from ropebwt2_amd.sharded import default_owners, rope_of
    for n in (2, 4, 8, 16):
        owner = default_owners(n)
        load = [0] * n
        for b in range(1, 5):
            for x in range(1, 5):
                load[owner[rope_of(b, x)]] += 1
        assert max(load) == -(-16 // n) and min(load) == 16 // n


@pytest.mark.parametrize("n", [2, 3])
def test_driver_over_gloo_cpu(n):
    p = launch(n, ["mock"], 29600 + n)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count("mock exchange ok") == n, out


@pytest.mark.gpu
@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("n", [2, 4])
def test_virtual_ranks_match_oracle(hip, so, n):
    from ropebwt2_amd.sharded import VirtualCluster
    reads = H.repetitive_reads(3000, seed=60 + so, genome_len=800, max_len=100)
    codes = H.splitmix_bases(4000, 75, seed=3)
    batches = [H.encode_batch(reads[:1800]), H.encode_batch_fixed(codes), H.encode_batch(reads[1800:], True, True)]
    o = H.Oracle(so)
    vc = VirtualCluster(so, n)
    for buf in batches:
        o.insert_multi(buf)
        vc.insert_multi(buf)
        assert np.array_equal(vc.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d" % b
    vc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [6, 8, 18])
def test_virtual_ranks_many(hip, n):
    """8 ranks: two heavy pieces each; 18 ranks: ranks >= 16 own nothing but take part in every collective"""
    from ropebwt2_amd.sharded import VirtualCluster
    codes = H.splitmix_bases(6000, 50, seed=21)
    reads = H.repetitive_reads(1200, seed=33)
    o = H.Oracle(2)
    vc = VirtualCluster(2, n)
    for buf in (H.encode_batch_fixed(codes, True, True), H.encode_batch(reads)):
        o.insert_multi(buf)
        vc.insert_multi(buf)
    assert np.array_equal(vc.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d" % b
    vc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_virtual_ranks_random_owner_maps(hip, seed):
    """any assignment of the 31 sub-ropes to ranks gives the same BWT (pieces of one rope scattered over ranks,
    ranks that own only light pieces, a rank that owns nothing)"""
    from ropebwt2_amd.sharded import NR, VirtualCluster
    rng = np.random.RandomState(seed)
    n = 5
    owners = [int(x) for x in rng.randint(0, n - 1, size=NR)]          # rank n-1 owns nothing
    so = seed % 3
    reads = H.repetitive_reads(2000, seed=70 + seed, genome_len=600, max_len=70)
    codes = H.splitmix_bases(3000, 64, seed=9 + seed)
    o = H.Oracle(so)
    vc = VirtualCluster(so, n, owners=owners)
    for buf in (H.encode_batch(reads[:1200]), H.encode_batch_fixed(codes), H.encode_batch(reads[1200:], True, True)):
        o.insert_multi(buf)
        vc.insert_multi(buf)
    assert np.array_equal(vc.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d" % b
    vc.close()


@pytest.mark.gpu
def test_virtual_ranks_golden(hip, golden):
    from ropebwt2_amd.sharded import VirtualCluster
    g = golden["sets"]["100k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    vc = VirtualCluster(1, 4)
    vc.insert_multi(H.encode_batch_fixed(codes[:60000]))
    vc.insert_multi(H.encode_batch_fixed(codes[60000:]))
    bwt = np.concatenate([vc.rope(b) for b in range(6)])
    assert H.md5(H.bwt_text(bwt) + b"\n") == g["text_md5"]["-LRs"]
    vc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("so", [0, 2])
def test_two_processes_over_gloo(hip, so):
    p = launch(2, ["gpu", so], 29650 + so)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count(" ok") >= 2, out


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="the RCCL path needs >= 2 GPUs on the box (one process per GPU)")
@pytest.mark.parametrize("so", [0, 2])
def test_processes_over_rccl(hip, so):
    """one process per GPU, backend nccl (= RCCL): all_reduce of the count matrix and all_to_all_single of the string
    records on device tensors; every rank compares its owned pieces with the oracle"""
    n = min(_gpu_count(), 8)
    p = launch(n, ["nccl", so], 29700 + so)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count("backend nccl") >= n and out.count(" ok") >= n, out


@pytest.mark.gpu
@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("n", [2, 8])
def test_stream_ordered_protocol(hip, so, n):
    """the protocol TorchComm uses over RCCL (engine on the caller's stream, count matrix reduced in place in a device tensor, no
    host synchronisation around merge / exchange / unpack) with N engines on one device and torch ops as collectives"""
    from ropebwt2_amd.sharded import StreamOrderedCluster
    reads = H.repetitive_reads(3000, seed=160 + so, genome_len=800, max_len=100)
    codes = H.splitmix_bases(4000, 75, seed=13)
    batches = [H.encode_batch(reads[:1800]), H.encode_batch_fixed(codes), H.encode_batch(reads[1800:], True, True)]
    o = H.Oracle(so)
    vc = StreamOrderedCluster(so, n)
    for buf in batches:
        o.insert_multi(buf)
        vc.insert_multi(buf)
        assert np.array_equal(vc.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d" % b
    vc.close()


@pytest.mark.gpu
def test_stream_ordered_protocol_golden_1M(hip, golden):
    from ropebwt2_amd.sharded import StreamOrderedCluster
    g = golden["sets"]["1M_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    vc = StreamOrderedCluster(2, 8)
    vc.insert_multi(H.encode_batch_fixed(codes[:600000], True, True))
    vc.insert_multi(H.encode_batch_fixed(codes[600000:], True, True))
    bwt = np.concatenate([vc.rope(b) for b in range(6)])
    assert H.md5(H.bwt_text(bwt) + b"\n") == g["text_md5"]["-Lr"]
    vc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_virtual_ranks_continue_a_loaded_index(hip, n):
    """configs[4] on a sharded index: every rank is handed the run bytes of all six ropes (what mr_restore_runs holds), the device
    loader keeps the pieces the rank owns and only counts the others; more batches through the sharded protocol; ropes = oracle"""
    from ropebwt2_amd.sharded import VirtualCluster
    from ropebwt2_amd.hipbwt import encode_runs
    reads = H.repetitive_reads(2400, seed=71, genome_len=600, max_len=90) + [[1] * 400] * 30
    for so in (0, 2):
        o = H.Oracle(so)
        o.insert_multi(H.encode_batch(reads[:1300]))
        vc = VirtualCluster(so, n)
        rles = [encode_runs(o.rope(b)) for b in range(6)]
        for r in vc.ranks:
            r.load_ropes(rles)
        assert np.array_equal(vc.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d after load (so %d)" % (b, so)
        for buf in (H.encode_batch(reads[1300:2000]), H.encode_batch(reads[2000:], True, so == 2)):
            o.insert_multi(buf)
            vc.insert_multi(buf)
            assert np.array_equal(vc.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(vc.rope(b), o.rope(b)), "rope %d (so %d)" % (b, so)
        vc.close()
