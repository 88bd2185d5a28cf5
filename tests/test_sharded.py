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


def test_exchange_layout_is_consistent():
    from ropebwt2_amd.sharded import default_owners, exchange_layout
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 4, 8):
        owner = default_owners(n)
        assert len(owner) == 6 and max(owner) < n
        g = rng.randint(0, 1000, size=(6, 6))
        sent = np.array([exchange_layout(owner, n, s, g) for s in range(n)])
        # everything that inserts a symbol 1..5 is sent exactly once, to the owner of that rope
        assert sent.sum() == g[:, 1:].sum()
        for d in range(n):
            assert sent[:, d].sum() == sum(g[:, a].sum() for a in range(1, 6) if owner[a] == d)


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
@pytest.mark.parametrize("n", [6, 8])
def test_virtual_ranks_with_idle_ranks(hip, n):
    """more ranks than ropes with load: ranks >= 4 own nothing but take part in every collective"""
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
