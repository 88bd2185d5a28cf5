"""Worker for the multi-process tests of ropebwt2_amd.sharded (launched by torch.distributed.run).

mode "gpu":  every rank drives a real HIP engine on cuda:0 (several processes share the one GPU of the
             test box), exchange over gloo with host staging; owned ropes are compared with the oracle.
mode "nccl": the same check with one GPU per rank and the exchange on device tensors over RCCL (backend "nccl"); needs as many
             GPUs as ranks -- tests/test_sharded.py runs it whenever torch.cuda.device_count() >= 2.
mode "crank": one process per GPU, the round loop INSIDE librb2hip.so (MultiBwt(rank=...) = rb2_hip_multi_create_rank): RCCL's C API,
             the ncclUniqueId handed round through torch.distributed; every rank compares the pieces it holds with the oracle.
mode "mock": no GPU: a toy engine emits tagged records following a random count matrix; checks that
             TorchComm delivers them exactly where rb2_hip_shard_finish's layout expects them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch.distributed as dist
    mode = sys.argv[1]
    dev = 0
    if mode in ("nccl", "crank"):
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    from ropebwt2_amd import sharded
    if mode == "crank":
        import helpers as H
        from ropebwt2_amd import MultiBwt
        so = int(sys.argv[2])
        box = [MultiBwt.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        m = MultiBwt(so, [dev], "rccl", rank=rank, nranks=n, nccl_id=box[0])
        owner = MultiBwt.default_owners(n)
        reads = H.repetitive_reads(2500, seed=90 + so, genome_len=700, max_len=90)
        codes = H.splitmix_bases(3000, 60, seed=5)
        o = H.Oracle(so)
        for buf in (H.encode_batch(reads[:1500]), H.encode_batch_fixed(codes), H.encode_batch(reads[1500:], True, True)):
            o.insert_multi(buf)
            m.insert_multi(buf)
            assert np.array_equal(m.counts(), o.counts()), "rank %d: count matrix differs" % rank
        c = o.counts()
        for b in range(6):                                          # a multi-process handle returns the pieces of rope b it holds, in order
            want = []
            for r in range(sharded.NR):
                if sharded.rope_sym(r) != b or owner[r] != rank:
                    continue
                x = sharded.rope_prev(r)
                lo = int(c[:x, b].sum()) if b else 0
                k = int(c[x, b]) if b else int(c[0].sum())
                want.append(o.rope(b)[lo:lo + k])
            want = np.concatenate(want) if want else np.zeros(0, np.uint8)
            assert np.array_equal(m.rope(b), want), "rank %d: pieces of rope %d differ" % (rank, b)
        print("rank %d/%d so %d C-level RCCL driver ok (host syncs in rounds: %d of %d rounds)" % (rank, n, so, m.stats()["host_syncs_in_rounds"], m.stats()["rounds"]))
        m.close()
    elif mode in ("gpu", "nccl"):
        import helpers as H
        so = int(sys.argv[2])
        bwt = sharded.ShardedBwt(so, rank, n, device=dev)
        comm = sharded.TorchComm(bwt)
        reads = H.repetitive_reads(2500, seed=90 + so, genome_len=700, max_len=90)
        codes = H.splitmix_bases(3000, 60, seed=5)
        batches = [H.encode_batch(reads[:1500]), H.encode_batch_fixed(codes), H.encode_batch(reads[1500:], True, True)]
        o = H.Oracle(so)
        for buf in batches:
            o.insert_multi(buf)
            p = bwt.dev_alloc(len(buf) + 64)
            bwt.L.rb2_hip_memcpy(bwt.h, p, buf.ctypes.data, len(buf), 0)
            comm.insert_multi_dev(p, len(buf))
            bwt.dev_free(p)
            assert np.array_equal(bwt.counts(), o.counts()), "rank %d: count matrix differs" % rank
        c = o.counts()
        for r in range(sharded.NR):
            b, x = sharded.rope_sym(r), sharded.rope_prev(r)
            lo = int(c[:x, b].sum()) if b else 0                    # piece (b,x) = the b-symbols of rope x
            n = int(c[x, b]) if b else int(c[0].sum())
            want = o.rope(b)[lo:lo + n]
            if r in bwt.owned():
                assert np.array_equal(bwt.piece(r), want), "rank %d: piece %d differs" % (rank, r)
            else:
                assert len(bwt.piece(r)) == 0
        print("rank %d/%d so %d backend %s owned %s ok" % (rank, n, so, dist.get_backend(), bwt.owned()))
    else:
        owner = sharded.default_owners(n)
        NR, sym, prev = sharded.NR, sharded.rope_sym, sharded.rope_prev
        rng = np.random.RandomState(7)                      # same matrices on every rank
        mats = [rng.randint(0, 12, size=(NR, 6)).astype(np.int64) for _ in range(5)]

        def plan(src, dst, g):
            """records src sends to dst, in the engine's order: tagged (round-less) (r, a, index)"""
            out = []
            for r2 in range(1, NR):
                if owner[r2] != dst: continue
                a, b = sym(r2), prev(r2)
                for r in range(NR):
                    if sym(r) != b or owner[r] != src: continue
                    out += [(r, a, i) for i in range(int(g[r, a]))]
            return out

        class Toy:
            """stands in for ShardedBwt: host memory plays the device"""
            def __init__(self):
                self.rank, self.nranks, self.owner = rank, n, owner
                self.keep = []
            def dev_alloc(self, nb):
                a = np.zeros(nb, np.uint8); self.keep.append(a); return a.ctypes.data
            def dev_free(self, p):
                pass
            def stage_out(self, host_ptr, dev_ptr, nb):
                import ctypes; ctypes.memmove(host_ptr, dev_ptr, nb)
            def stage_in(self, dev_ptr, host_ptr, nb):
                import ctypes; ctypes.memmove(dev_ptr, host_ptr, nb)
            def batch_protocol(self, dev_ptr, nbytes, send_ptr_of, recv_ptr_of):
                import ctypes
                cap = NR * 6 * 12
                sp, rp = send_ptr_of(cap), recv_ptr_of(cap)
                for rnd, g in enumerate(mats):
                    loc = np.zeros((NR, 6), np.int64)
                    for r in range(NR):
                        if owner[r] == rank:
                            loc[r] = g[r]
                    tot = yield ("allreduce", loc.reshape(-1).copy())
                    assert np.array_equal(tot.reshape(NR, 6), g)
                    recs = [(rnd,) + t for d in range(n) for t in plan(rank, d, g)]
                    arr = np.zeros((len(recs), sharded.REC_BYTES // 8), np.int64)   # records: (round << 32 | r, a << 32 | index, 0...)
                    if recs: arr[:, :2] = [(t[0] << 32 | t[1], t[2] << 32 | t[3]) for t in recs]
                    raw = arr.view(np.uint8).reshape(-1)
                    ctypes.memmove(sp, raw.ctypes.data, len(raw)) if len(raw) else None
                    sc = sharded.exchange_layout(owner, n, rank, g)
                    rc = [sharded.exchange_layout(owner, n, s, g)[rank] for s in range(n)]
                    assert sc == [len(plan(rank, d, g)) for d in range(n)]
                    yield ("alltoall", sc, rc, int(g[:, 1:].sum()))
                    tot_r = sum(rc)
                    got = np.zeros((tot_r, sharded.REC_BYTES // 8), np.int64)
                    if tot_r: ctypes.memmove(got.ctypes.data, rp, tot_r * sharded.REC_BYTES)
                    want = [(rnd,) + t for s in range(n) for t in plan(s, rank, g)]
                    assert got[:, :2].tolist() == [[w[0] << 32 | w[1], w[2] << 32 | w[3]] for w in want], "round %d: records out of place" % rnd

        toy = Toy()
        comm = sharded.TorchComm(toy)
        comm.insert_multi_dev(0, 0)
        print("rank %d/%d mock exchange ok" % (rank, n))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
