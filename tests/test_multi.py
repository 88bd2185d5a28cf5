"""N GPUs behind ONE handle (include/rb2_hip.h rb2_hip_multi_*, csrc/rb2_multi.h): the sharded build driven inside the
library -- what mr_insert_multi does with its worker threads (/root/reference/mrope.c:287-296, 312-340) -- through the C ABI,
the mrope C API and the CLI (RB2_HIP_DEVICES).

CPU  : the owner map, the exchange plan against a simulation, and the exchange itself with world_size 2 and 3 over gloo.
GPU  : N virtual ranks on one device over the PEER transport (the complete round loop, device-side exchange plan, records
       fetched from the senders' buffers, no host synchronisation between rounds) bit-exact against the oracle and the
       reference's goldens; the RCCL transport on a group of one (librccl really loaded and called); a real multi-GPU run
       of both transports whenever the box has >= 2 GPUs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H


HERE = os.path.dirname(os.path.abspath(__file__))


def launch(n, args, port):
    """tests/multi_worker.py as n processes under torch.distributed.run"""
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "multi_worker.py")] + [str(a) for a in args]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)


def test_subrope_indexing():
    from helpers import NR, rope_sym, rope_prev, rope_of
    from ropebwt2_amd import load_hip_lib, build_all
    build_all()
    assert NR == load_hip_lib().rb2_hip_num_subropes() == 31 and rope_sym(0) == 0 and rope_prev(0) == 0
    seen = set()
    for b in range(1, 6):
        for x in range(6):
            r = rope_of(b, x)
            assert 1 <= r < NR and rope_sym(r) == b and rope_prev(r) == x
            seen.add(r)
    assert len(seen) == 30
    # pieces of one rope are contiguous and ordered by x: concatenating pieces in index order gives rope order
    assert [rope_sym(r) for r in range(NR)] == sorted(rope_sym(r) for r in range(NR))


def test_default_owners():
    """the library's owner map: equal to its Python restatement; every rank up to 16 carries load; on uniform DNA the heaviest
    rank holds at most ceil(16/n) of the 16 heavy pieces"""
    from ropebwt2_amd import MultiBwt, build_all
    from helpers import NR, default_owners, rope_of
    build_all()
    for n in (1, 2, 3, 4, 5, 7, 8, 16, 20, 64):
        owner = MultiBwt.default_owners(n)
        assert owner == default_owners(n) and len(owner) == NR and max(owner) < n
        if n <= 16:
            assert len(set(owner)) == n
    for n in (2, 4, 8, 16):
        owner = MultiBwt.default_owners(n)
        load = [0] * n
        for b in range(1, 5):
            for x in range(1, 5):
                load[owner[rope_of(b, x)]] += 1
        assert max(load) == -(-16 // n) and min(load) == 16 // n


def test_exchange_layout_is_consistent():
    from helpers import NR, default_owners, exchange_layout, rope_sym
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 4, 8, 16, 20):
        owner = default_owners(n)
        g = rng.randint(0, 1000, size=(NR, 6))
        sent = np.array([exchange_layout(owner, n, s, g) for s in range(n)])
        # everything that inserts a symbol 1..5 is sent exactly once, to the owner of piece (a, b)
        assert sent.sum() == g[:, 1:].sum()
        for d in range(n):
            want = 0
            for r in range(NR):
                for a in range(1, 6):
                    if owner[1 + (a - 1) * 6 + rope_sym(r)] == d:
                        want += g[r, a]
            assert sent[:, d].sum() == want


@pytest.mark.parametrize("n", [2, 3])
def test_exchange_over_gloo_cpu(n):
    """world_size 2 and 3 on the CPU: all_reduce of the count matrix, the library's exchange plan, the records through
    all_to_all_single -- every record where the plan of its receiver expects it (tests/multi_worker.py, mode "plan")"""
    from ropebwt2_amd import build_all
    build_all()
    p = launch(n, ["plan"], 29600 + n)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count("plan exchange ok") == n, out


def _plan(owner, n, g, me):
    """rb2_hip_multi_plan_host: the per-entry function of k_mround, run on the host"""
    from ropebwt2_amd import load_hip_lib
    L = load_hip_lib()
    own = (C.c_int * 31)(*owner)
    gg = np.ascontiguousarray(g, dtype=np.int64).reshape(-1)
    sd = np.zeros(31 * 6, np.int64)
    pcs = np.zeros((31 * 6, 5), np.int64)
    tot = C.c_int64(0)
    k = L.rb2_hip_multi_plan_host(own, n, gg.ctypes.data, me, sd.ctypes.data, pcs.ctypes.data, C.byref(tot))
    return sd.reshape(31, 6), pcs[:k], int(tot.value)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 16, 20])
def test_device_exchange_plan_against_a_simulated_exchange(n):
    """k_mround's plan (where every rank writes its records, which pieces every rank fetches from whom, where they land) against
    a literal simulation: the strings of bucket (a,b) of the next round are the members of the pieces (b,x), x = $ACGTN in order,
    that inserted a, in their old order (the stable scatter of mrope.c:303-309) -- for the default and for random owner maps"""
    from ropebwt2_amd import MultiBwt, build_all
    from helpers import NR, rope_sym, rope_prev, rope_of, exchange_layout
    build_all()
    rng = np.random.RandomState(n)
    for trial in range(6):
        owner = MultiBwt.default_owners(n) if trial < 2 else [int(x) for x in rng.randint(0, n, size=NR)]
        g = rng.randint(0, 9, size=(NR, 6)).astype(np.int64)
        g[rng.rand(NR, 6) < 0.3] = 0
        for r in range(NR):
            if rope_sym(r) == 0 and r != 0:
                g[r] = 0
        # every rank fills its send buffer: entry (r, a) at sdest, records tagged (r, a, i)
        send = []
        for s_ in range(n):
            sd, _, _ = _plan(owner, n, g, s_)
            per = exchange_layout(owner, n, s_, g)
            buf = [None] * sum(per)
            for r in range(NR):
                for a in range(1, 6):
                    if owner[r] == s_:
                        assert sd[r, a] >= 0
                        for i in range(int(g[r, a])):
                            assert buf[sd[r, a] + i] is None
                            buf[sd[r, a] + i] = (r, a, i)
                    else:
                        assert sd[r, a] == -1
            assert all(x is not None for x in buf)
            send.append(buf)
        for d in range(n):
            _, pcs, tot = _plan(owner, n, g, d)
            nxt, recv_order = {}, []
            for (src, off, vsrc, dst, cnt) in pcs.tolist():
                assert vsrc == len(recv_order)                        # pieces tile the receive order
                for i in range(cnt):
                    rec = send[src][off + i]
                    recv_order.append(rec)
                    assert dst + i not in nxt
                    nxt[dst + i] = rec
            assert tot == len(recv_order) == len(nxt)
            want = []                                                 # next arrays of rank d: its pieces ascending; inside (a,b): sources (b,x) by x, old order
            for r2 in range(1, NR):
                if owner[r2] != d:
                    continue
                a, b = rope_sym(r2), rope_prev(r2)
                for r in range(NR):
                    if rope_sym(r) == b:
                        want += [(r, a, i) for i in range(int(g[r, a]))]
            assert [nxt[i] for i in range(len(want))] == want and len(nxt) == len(want)
            # the RCCL receive buffer: blocks by source rank, each in the sender's order for this destination
            blocks = []
            for s_ in range(n):
                per = exchange_layout(owner, n, s_, g)
                st = sum(per[:d])
                blocks += send[s_][st:st + per[d]]
            assert recv_order == blocks


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


def _batches(so):
    reads = H.repetitive_reads(3000, seed=60 + so, genome_len=800, max_len=100)
    codes = H.splitmix_bases(4000, 75, seed=3)
    return [H.encode_batch(reads[:1800]), H.encode_batch_fixed(codes), H.encode_batch(reads[1800:], True, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_peer_virtual_ranks_match_oracle(hip, so, n):
    from ropebwt2_amd import MultiBwt
    o = H.Oracle(so)
    m = MultiBwt(so, [0] * n, "peer")
    for buf in _batches(so):
        o.insert_multi(buf)
        m.insert_multi(buf)
        assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    st = m.stats()
    assert st["host_syncs_in_rounds"] == 0 and st["batches"] == 3 and st["rounds"] > 100
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 6, 18])
def test_peer_many_ranks_and_odd_counts(hip, n):
    """18 ranks: ranks >= 16 own nothing but take part in every round; 3 and 6: unbalanced owner maps"""
    from ropebwt2_amd import MultiBwt
    codes = H.splitmix_bases(6000, 50, seed=21)
    reads = H.repetitive_reads(1200, seed=33)
    o = H.Oracle(2)
    m = MultiBwt(2, [0] * n, "peer")
    for buf in (H.encode_batch_fixed(codes, True, True), H.encode_batch(reads)):
        o.insert_multi(buf)
        m.insert_multi(buf)
    assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_peer_random_owner_maps(hip, seed):
    """any assignment of the 31 sub-ropes to ranks gives the same BWT: the device-side exchange plan (k_mround) against
    pieces of one rope scattered over ranks, ranks with only light pieces, a rank that owns nothing"""
    from ropebwt2_amd import MultiBwt
    rng = np.random.RandomState(seed)
    n = 5
    owners = [int(x) for x in rng.randint(0, n - 1, size=31)]
    so = seed % 3
    reads = H.repetitive_reads(2000, seed=70 + seed, genome_len=600, max_len=70)
    codes = H.splitmix_bases(3000, 64, seed=9 + seed)
    o = H.Oracle(so)
    m = MultiBwt(so, [0] * n, "peer", owners=owners)
    for buf in (H.encode_batch(reads[:1200]), H.encode_batch_fixed(codes), H.encode_batch(reads[1200:], True, True)):
        o.insert_multi(buf)
        m.insert_multi(buf)
    assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_peer_fuzz_jobs(hip, seed):
    """random jobs: number of ranks, owner map, order, strands, read shapes, batch cuts, and the leaf layout forced sparse in half
    of them -- ropes and count matrix against the oracle after every batch"""
    from ropebwt2_amd import MultiBwt
    rng = np.random.RandomState(1000 + seed)
    n = int(rng.randint(1, 10))
    owners = None if rng.rand() < 0.5 else [int(x) for x in rng.randint(0, n, size=31)]
    so = int(rng.randint(0, 3))
    reads = []
    for _ in range(int(rng.randint(2, 5))):
        kind = rng.randint(0, 3)
        if kind == 0:
            reads += H.repetitive_reads(int(rng.randint(50, 1500)), seed=int(rng.randint(1 << 30)), genome_len=int(rng.randint(60, 2000)), max_len=int(rng.randint(5, 300)))
        elif kind == 1:
            reads += list(H.splitmix_bases(int(rng.randint(10, 3000)), int(rng.randint(1, 250)), seed=int(rng.randint(1 << 30))))
        else:
            reads += [[int(rng.randint(1, 5))] * int(rng.randint(1, 3000))] * int(rng.randint(1, 40))     # homopolymers, duplicates
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    cuts = sorted(set([0, len(reads)] + [int(x) for x in rng.randint(0, len(reads) + 1, size=int(rng.randint(0, 4)))]))
    env = {"RB2_SPARSE_LAMBDA": "1e18", "RB2_SPARSE_MAXPEN": "0"} if seed % 2 else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        o = H.Oracle(so)
        m = MultiBwt(so, [0] * n, "peer", owners=owners)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if a == b:
                continue
            buf = H.encode_batch(reads[a:b], True, bool(rng.rand() < 0.4))
            o.insert_multi(buf)
            m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts()), "counts (seed %d, %d ranks)" % (seed, n)
        for b in range(6):
            assert np.array_equal(m.rope(b), o.rope(b)), "rope %d (seed %d, %d ranks, so %d)" % (b, seed, n, so)
        m.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_peer_edge_batches(hip):
    """empty strings, one string, a batch of sentinels only, strings of very different lengths"""
    from ropebwt2_amd import MultiBwt
    for so in (0, 1, 2):
        o = H.Oracle(so)
        m = MultiBwt(so, [0] * 4, "peer")
        bufs = [np.zeros(5, np.uint8), H.encode_batch([[1, 2, 3, 4, 5, 1]]), H.encode_batch([[], [1], [], [2, 2, 2, 2] * 200, [4, 3]]),
                H.encode_batch(H.repetitive_reads(300, seed=5), True, True)]
        for buf in bufs:
            o.insert_multi(buf)
            m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(m.rope(b), o.rope(b)), "rope %d (so %d)" % (b, so)
        m.close()


@pytest.mark.gpu
def test_peer_golden_1M_rclo_both_strands(hip, golden):
    from ropebwt2_amd import MultiBwt
    g = golden["sets"]["1M_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    m = MultiBwt(2, [0] * 8, "peer")
    m.insert_multi(H.encode_batch_fixed(codes[:600000], True, True))
    m.insert_multi(H.encode_batch_fixed(codes[600000:], True, True))
    bwt = np.concatenate([m.rope(b) for b in range(6)])
    assert H.md5(H.bwt_text(bwt) + b"\n") == g["text_md5"]["-Lr"]
    assert m.stats()["host_syncs_in_rounds"] == 0
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_peer_continue_a_loaded_index(hip, n):
    """configs[4] on a sharded index through the one-handle API: load_ropes (every rank keeps its pieces), more batches, rank queries"""
    from ropebwt2_amd import MultiBwt
    from ropebwt2_amd.hipbwt import encode_runs
    reads = H.repetitive_reads(2400, seed=71, genome_len=600, max_len=90) + [[1] * 400] * 30
    for so in (0, 2):
        o = H.Oracle(so)
        o.insert_multi(H.encode_batch(reads[:1300]))
        m = MultiBwt(so, [0] * n, "peer")
        m.load_ropes([encode_runs(o.rope(b)) for b in range(6)])
        assert np.array_equal(m.counts(), o.counts())
        for buf in (H.encode_batch(reads[1300:2000]), H.encode_batch(reads[2000:], True, so == 2)):
            o.insert_multi(buf)
            m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts())
        for b in range(6):
            ro = o.rope(b)
            assert np.array_equal(m.rope(b), ro), "rope %d (so %d)" % (b, so)
            for x in (0, 1, len(ro) // 3, len(ro) - 1, len(ro)):
                if 0 <= x <= len(ro):
                    assert np.array_equal(m.rank1a(b, x), np.bincount(ro[:x], minlength=6)), (b, x)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("so", [0, 2])
def test_rccl_group_of_one(hip, so):
    """the RCCL transport on a one-rank communicator: librccl is loaded, ncclAllReduce reduces the matrix in place, the rank's
    own block travels through grouped ncclSend / ncclRecv (RB2_RCCL_SELF=1) -- every call of the multi-GPU transport, on one GPU"""
    from ropebwt2_amd import MultiBwt
    for self_msgs in ("1", "0"):
        os.environ["RB2_RCCL_SELF"] = self_msgs
        try:
            o = H.Oracle(so)
            m = MultiBwt(so, [0], "rccl")
            for buf in _batches(so):
                o.insert_multi(buf)
                m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts())
            for b in range(6):
                assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
            assert m.stats()["host_syncs_in_rounds"] == m.stats()["rounds"]     # one event wait per round, behind nothing but the reduce
            m.close()
        finally:
            del os.environ["RB2_RCCL_SELF"]


@pytest.mark.gpu
@pytest.mark.parametrize("transport,n", [("peer", 4), ("peer", 8), ("rccl", 1)])
def test_start_up_self_test(hip, transport, n):
    """rb2_hip_multi_create's self-test (on by itself whenever the ranks sit on more than one physical device; forced here): a
    2000-read job across the ranks against the same job on one engine, sub-rope checksums + count matrix; the handle comes back
    empty, with clean statistics, and builds the right index afterwards"""
    from ropebwt2_amd import MultiBwt
    os.environ["RB2_MULTI_SELFTEST"] = "1"
    os.environ["RB2_HIP_TRACE"] = "0"
    try:
        m = MultiBwt(1, [0] * n, transport)
    finally:
        del os.environ["RB2_MULTI_SELFTEST"]; del os.environ["RB2_HIP_TRACE"]
    assert m.L.rb2_hip_multi_transport(m.h) == {"peer": 0, "rccl": 1}[transport]
    st = m.stats()
    assert st["rounds"] == 0 and st["batches"] == 0 and int(m.counts().sum()) == 0
    o = H.Oracle(1)
    for buf in _batches(1)[:2]:
        o.insert_multi(buf); m.insert_multi(buf)
    assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    m.close()


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs >= 2 GPUs on the box")
@pytest.mark.parametrize("transport", ["peer", "rccl"])
def test_real_devices_one_process(hip, transport):
    """one process, one rank per visible GPU: peer access over xGMI / RCCL inside a process"""
    from ropebwt2_amd import MultiBwt
    n = min(_gpu_count(), 8)
    for so in (0, 2):
        o = H.Oracle(so)
        m = MultiBwt(so, list(range(n)), transport)
        for buf in _batches(so):
            o.insert_multi(buf)
            m.insert_multi(buf)
        assert np.array_equal(m.counts(), o.counts())
        for b in range(6):
            assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
        m.close()


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs >= 2 GPUs on the box (one process per GPU)")
@pytest.mark.parametrize("so", [0, 2])
def test_processes_over_rccl_c_driver(hip, so):
    """one process per GPU, each ONE rank of an RCCL group driven inside librb2hip.so (rb2_hip_multi_create_rank): what
    bench.py runs under torch.distributed.run"""
    n = min(_gpu_count(), 8)
    p = launch(n, ["crank", so], 29750 + so)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count("C-level RCCL driver ok") >= n, out


@pytest.mark.gpu
def test_one_process_group_over_rccl_c_driver(hip):
    """the same worker as a group of ONE process on the one-GPU box: unique id through torch.distributed, ncclCommInitRank,
    the round loop with ncclAllReduce on the real communicator"""
    p = launch(1, ["crank", 2], 29760)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert "C-level RCCL driver ok" in out, out


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_with_one_process(hip):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per GPU, RCCL group driven inside the
    library, a fresh ncclUniqueId per communicator: warm-up handle and measured handle), with N = 1 forced onto that path"""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RB2_BENCH_FORCE_MULTI="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29771",
           os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--reads", "3000000", "--batch", "0.1", "--no-extras", "--no-cpu-baseline"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["config"]["counts_ok"] is True and "RCCL C API" in d["config"]["driver"], d["config"]
    assert d["config"]["multi_stats"]["rounds"] > 100 and d["value"] > 0


# ---- through the drop-in boundary: the CLI and the mrope C API with RB2_HIP_DEVICES ---------------------------------------

def _cli_env(flags, data, devices, transport=None, extra_env=None):
    from test_host_layer import CLI
    env = dict(os.environ, RB2_HIP_DEVICES=devices)
    if transport:
        env["RB2_HIP_TRANSPORT"] = transport
    env.update(extra_env or {})
    p = subprocess.run([CLI] + list(flags) + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("flag", ["-LR", "-LRs", "-LRr", "-L", "-Lr", "-LRN"])
def test_cli_kat_on_8_ranks(hip, golden, flag):
    assert _cli_env([flag], golden["kat_input"].encode(), "0,0,0,0,0,0,0,0").decode().strip() == golden["kat"][flag]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["10k_x_101", "1M_x_101"])
@pytest.mark.parametrize("flag", ["-LRrd", "-Lrd", "-LRsd"])
def test_cli_goldens_on_8_ranks(hip, golden, name, flag):
    """`ropebwt2 -brR`-style builds with the index sharded over 8 ranks behind mr_insert_multi: the reference's .fmd bytes"""
    g = golden["sets"][name]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    extra = ["-m20m"] if name == "1M_x_101" else []                  # several batches onto the sharded index
    assert H.md5(_cli_env([flag] + extra, text, "0,0,0,0,0,0,0,0")) == g["fmd_md5"][flag]


@pytest.mark.gpu
@pytest.mark.parametrize("so_flag", ["", "s", "r"])
def test_cli_incremental_on_8_ranks(hip, golden, so_flag, tmp_path):
    """configs[4] shape (-bi) through the CLI on a sharded index: .fmr of the first half (one GPU) + second half on 8 ranks"""
    from test_host_layer import cli
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    half = tmp_path / "half.fmr"
    half.write_bytes(cli(["-LRb" + so_flag], H.reads_to_text(codes[:5000])))
    out = _cli_env(["-LRd", "-i", str(half)], H.reads_to_text(codes[5000:]), "0,0,0,0,0,0,0,0")
    assert H.md5(out) == g["fmd_md5"]["-LR" + so_flag + "d"]
    # and the other way round: the sharded build's .fmr continues on one GPU
    half.write_bytes(_cli_env(["-LRb" + so_flag], H.reads_to_text(codes[:5000]), "0,0,0,0"))
    assert H.md5(cli(["-LRd", "-i", str(half)], H.reads_to_text(codes[5000:]))) == g["fmd_md5"]["-LR" + so_flag + "d"]


@pytest.mark.gpu
def test_cli_rccl_transport_group_of_one(hip, golden):
    """RB2_HIP_DEVICES with ONE device selects the single engine; two entries + rccl on one device must refuse loudly"""
    from test_host_layer import CLI
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    assert H.md5(_cli_env(["-LRsd"], text, "0")) == g["fmd_md5"]["-LRsd"]
    p = subprocess.run([CLI, "-LRsd", "-"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, RB2_HIP_DEVICES="0,0", RB2_HIP_TRANSPORT="rccl"))
    assert p.returncode != 0 and b"one device per rank" in p.stderr


@pytest.mark.gpu
def test_mrope_c_api_on_4_ranks(hip):
    """mr_init / mr_insert_multi / mr_rank2a / mr_itr_* of libropebwt2.so with the index sharded over four ranks"""
    code = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers as H
from ropebwt2_amd.build import lib_path
L = C.CDLL(lib_path("libropebwt2.so"))
L.mr_init.restype = C.c_void_p; L.mr_init.argtypes = [C.c_int, C.c_int, C.c_int]
L.mr_insert_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
L.mr_rank2a.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
L.mr_destroy.argtypes = [C.c_void_p]
L.mr_hip_multi_handle.restype = C.c_void_p; L.mr_hip_multi_handle.argtypes = [C.c_void_p]
L.mr_hip_handle.restype = C.c_void_p; L.mr_hip_handle.argtypes = [C.c_void_p]
for so in (0, 1, 2):
    o = H.Oracle(so)
    mr = L.mr_init(64, 512, so)
    reads = H.repetitive_reads(1500, seed=11 + so, genome_len=500, max_len=80)
    for buf in (H.encode_batch(reads[:900]), H.encode_batch(reads[900:], True, True)):
        o.insert_multi(buf)
        L.mr_insert_multi(mr, len(buf), buf.ctypes.data, 1)
    assert L.mr_hip_multi_handle(mr) and not L.mr_hip_handle(mr)
    bwt = o.bwt()
    for x in (0, 1, 17, len(bwt) // 2, len(bwt) - 1, len(bwt)):
        cx = (C.c_int64 * 6)()
        L.mr_rank2a(mr, x, -1, cx, None)
        assert list(cx) == list(np.bincount(bwt[:x], minlength=6)), (so, x)
    L.mr_destroy(mr)
print("c api ok")
''' % (H.ROOT, os.path.join(H.ROOT, "tests"))
    import sys
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RB2_HIP_DEVICES="0,0,0,0"))
    assert p.returncode == 0 and b"c api ok" in p.stdout, p.stderr.decode()[-2000:]


@pytest.mark.gpu
def test_fatal_error_on_a_rank_thread_reaches_the_handler_on_the_calling_thread(hip):
    """N ranks behind one handle: a failure on a rank's own host thread (here: a batch with a byte that is no nt6 code, found by the
    rank that holds rope $) is reported through rb2_hip_set_fatal_handler on the thread that called the API, after the rank threads are joined"""
    import sys
    root = os.path.dirname(HERE)
    code = ("import sys, os, threading, ctypes as C; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from ropebwt2_amd.hipbwt import load_hip_lib\n"
            "L = load_hip_lib()\n"
            "CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)\n"
            "main = threading.get_ident()\n"
            "def h(user, msg):\n"
            "    sys.stdout.write('handler on %%s thread: ' %% ('the calling' if threading.get_ident() == main else 'another') + msg.decode()); sys.stdout.flush(); os._exit(7)\n"
            "cb = CB(h)\n"
            "L.rb2_hip_set_fatal_handler(cb, None)\n"
            "dev = (C.c_int * 4)(0, 0, 0, 0)\n"
            "L.rb2_hip_multi_create.restype = C.c_void_p\n"
            "m = L.rb2_hip_multi_create(4, dev, 0, 0, None)\n"
            "s = np.array([1, 2, 9, 4, 0, 3, 3, 0], np.uint8)\n"
            "L.rb2_hip_multi_insert_multi(C.c_void_p(m), C.c_int64(len(s)), s.ctypes.data_as(C.c_void_p))\n") % root
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 7, (p.returncode, p.stderr.decode()[-400:])
    assert b"handler on the calling thread: [rb2_hip] the batch contains bytes that are not nt6 codes" in p.stdout, p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_peer_shared_batch_text(hip, n, monkeypatch):
    """Ranks of one process (PEER): the text of a host-buffer batch is ONE copy, its pieces spread over the ranks' devices and mapped
    for all of them (RB2_MULTI_TEXT=shard forces that on one device) -- every rank holds about 1 / n of it, results as ever."""
    from ropebwt2_amd import MultiBwt
    monkeypatch.setenv("RB2_MULTI_TEXT", "shard")
    so = 1
    o = H.Oracle(so)
    m = MultiBwt(so, [0] * n, "peer")
    for buf in _batches(so):                                     # small batches: one 64 MiB piece, held by rank 0
        o.insert_multi(buf)
        m.insert_multi(buf)
        tb = m.text_bytes()
        assert tb[0] == 64 << 20 and sum(tb[1:]) == 0, tb
        assert np.array_equal(m.counts(), o.counts())
    for b in range(6):
        assert np.array_equal(m.rope(b), o.rope(b)), "rope %d" % b
    m.close()
    # a batch of several pieces against one engine (device-side checksums of the ropes + the count matrix)
    L, reads = 100, 3_000_000
    one = hip.HipBwt(so, 0)
    buf = one.dev_alloc(reads * (L + 1) + 64)
    one.synth_reads(buf, 0, reads, L, seed=77)
    one.sync()
    host = np.empty(reads * (L + 1), np.uint8)
    one.L.rb2_hip_memcpy(one.h, host.ctypes.data, buf, host.size, 1)
    one.sync()
    one.insert_multi_dev(buf, reads * (L + 1))
    one.sync()
    m = MultiBwt(so, [0] * n, "peer")
    m.insert_multi(host)
    tb = m.text_bytes()
    need = reads * (L + 1) + 64
    assert sum(tb) >= need and sum(tb) < need + 2 * max(tb) and max(tb) <= -(-need // n) + (128 << 20), (tb, need)
    assert sum(1 for x in tb if x > 0) >= min(n, 2)
    assert np.array_equal(m.counts(), one.counts())
    assert m.rope_hashes() == one.rope_hashes()
    monkeypatch.setenv("RB2_MULTI_TEXT", "copy")                # ... and switched off: the whole batch on the first rank of the device
    m2 = MultiBwt(so, [0] * n, "peer")
    m2.insert_multi(host)
    tb2 = m2.text_bytes()
    assert tb2[0] >= need and sum(tb2[1:]) == 0, tb2
    assert m2.rope_hashes() == one.rope_hashes()
    one.dev_free(buf); one.close(); m.close(); m2.close()
