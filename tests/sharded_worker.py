"""Worker for the multi-process tests of ropebwt2_amd.sharded (launched by torch.distributed.run).

mode "gpu":  every rank drives a real HIP engine on cuda:0 (several processes share the one GPU of the
             test box), exchange over gloo with host staging; owned ropes are compared with the oracle.
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
    dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    from ropebwt2_amd import sharded
    if mode == "gpu":
        import helpers as H
        so = int(sys.argv[2])
        bwt = sharded.ShardedBwt(so, rank, n, device=0)
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
        for b in bwt.owned():
            assert np.array_equal(bwt.rope(b), o.rope(b)), "rank %d: rope %d differs" % (rank, b)
        for b in range(6):
            if b not in bwt.owned():
                assert len(bwt.rope_rle(b)) == 0
        print("rank %d/%d so %d owned %s ok" % (rank, n, so, bwt.owned()))
    else:
        owner = sharded.default_owners(n)
        rng = np.random.RandomState(7)                      # same matrices on every rank
        mats = [rng.randint(0, 40, size=(6, 6)).astype(np.int64) for _ in range(5)]

        class Toy:
            """stands in for ShardedBwt: host memory plays the device"""
            def __init__(self):
                self.rank, self.nranks, self.owner = rank, n, owner
                self.keep = []
                self.received = []
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
                cap = 6 * 6 * 40 * n
                sp, rp = send_ptr_of(cap), recv_ptr_of(cap)
                for r, g in enumerate(mats):
                    loc = np.zeros((6, 6), np.int64)
                    for b in range(6):
                        if owner[b] == rank:
                            loc[b] = g[b]
                    tot = yield ("allreduce", loc.reshape(-1).copy())
                    assert np.array_equal(tot.reshape(6, 6), g)
                    # records tagged (round, b, a, index), written in the engine's send layout
                    recs = []
                    for d in range(n):
                        for a in range(1, 6):
                            if owner[a] != d: continue
                            for b in range(6):
                                if owner[b] != rank: continue
                                for i in range(int(g[b, a])):
                                    recs.append((r, b, a, i))
                    arr = np.zeros((len(recs), 4), np.int64)
                    if recs: arr[:] = recs
                    raw = arr.view(np.uint8).reshape(-1)
                    ctypes.memmove(sp, raw.ctypes.data, len(raw)) if len(raw) else None
                    sc = sharded.exchange_layout(owner, n, rank, g)
                    rc = [sharded.exchange_layout(owner, n, s, g)[rank] for s in range(n)]
                    yield ("alltoall", sc, rc)
                    tot_r = sum(rc)
                    got = np.zeros((tot_r, 4), np.int64)
                    if tot_r: ctypes.memmove(got.ctypes.data, rp, tot_r * 32)
                    want = []
                    for s in range(n):
                        for a in range(1, 6):
                            if owner[a] != rank: continue
                            for b in range(6):
                                if owner[b] != s: continue
                                want += [(r, b, a, i) for i in range(int(g[b, a]))]
                    assert got.tolist() == [list(w) for w in want], "round %d: records out of place" % r

        toy = Toy()
        comm = sharded.TorchComm(toy)
        comm.insert_multi_dev(0, 0)
        print("rank %d/%d mock exchange ok" % (rank, n))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
