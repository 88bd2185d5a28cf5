"""Worker for the multi-process tests of the sharded build (launched by torch.distributed.run from tests/test_multi.py).

mode "crank": one process per GPU, the round loop INSIDE librb2hip.so (MultiBwt(rank=...) = rb2_hip_multi_create_rank): RCCL's C API,
             the ncclUniqueId handed round through torch.distributed; every rank compares the pieces it holds with the oracle.
mode "plan":  no GPU, gloo: the count matrix is summed by all_reduce, the exchange plan comes from the library
             (rb2_hip_multi_plan_host = the device's k_mround on the host), tagged records travel through all_to_all_single and
             must land where the plan says -- the RCCL transport's layout on real processes.
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
    if mode == "crank":
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
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
            for r in range(H.NR):
                if H.rope_sym(r) != b or owner[r] != rank:
                    continue
                x = H.rope_prev(r)
                lo = int(c[:x, b].sum()) if b else 0
                k = int(c[x, b]) if b else int(c[0].sum())
                want.append(o.rope(b)[lo:lo + k])
            want = np.concatenate(want) if want else np.zeros(0, np.uint8)
            assert np.array_equal(m.rope(b), want), "rank %d: pieces of rope %d differ" % (rank, b)
        print("rank %d/%d so %d C-level RCCL driver ok (host syncs in rounds: %d of %d rounds)" % (rank, n, so, m.stats()["host_syncs_in_rounds"], m.stats()["rounds"]))
        m.close()
    else:
        # "plan": no GPU.  Five rounds of the exchange as the RCCL transport lays it out (rb2_multi.h): every rank knows only the
        # rows of the count matrix of the pieces it owns -> all_reduce; rb2_hip_multi_plan_host (the per-entry function of the
        # device's k_mround) says where this rank writes the records it sends and which pieces it receives from whom; the records
        # travel through all_to_all_single over gloo and must arrive exactly where the plan expects them.
        import ctypes as C
        import helpers as H
        import torch
        from ropebwt2_amd import MultiBwt, load_hip_lib
        L = load_hip_lib()
        NR = H.NR
        rng = np.random.RandomState(7)                      # the same matrices and owner maps on every rank
        for rnd in range(5):
            owner = MultiBwt.default_owners(n) if rnd < 3 else [int(x) for x in rng.randint(0, n, size=NR)]
            assert rnd >= 3 or owner == H.default_owners(n)
            g = rng.randint(0, 12, size=(NR, 6)).astype(np.int64)
            g[rng.rand(NR, 6) < 0.25] = 0
            loc = np.zeros((NR, 6), np.int64)
            for r in range(NR):
                if owner[r] == rank:
                    loc[r] = g[r]
            t = torch.from_numpy(loc.reshape(-1).copy())
            dist.all_reduce(t)
            assert np.array_equal(t.numpy().reshape(NR, 6), g)
            own = (C.c_int * NR)(*owner)
            gg = np.ascontiguousarray(t.numpy())
            sd = np.zeros(NR * 6, np.int64); pcs = np.zeros((NR * 6, 5), np.int64); tot = C.c_int64(0)
            k = L.rb2_hip_multi_plan_host(own, n, gg.ctypes.data, rank, sd.ctypes.data, pcs.ctypes.data, C.byref(tot))
            sd = sd.reshape(NR, 6)
            sc = H.exchange_layout(owner, n, rank, g)
            rc = [H.exchange_layout(owner, n, s, g)[rank] for s in range(n)]
            send = np.full((max(1, sum(sc)), H.REC_WORDS), -1, np.int64)   # records tagged (round << 32 | r, a << 32 | index, 0)
            for r in range(NR):
                for a in range(1, 6):
                    if owner[r] == rank:
                        for i in range(int(g[r, a])):
                            assert send[sd[r, a] + i, 0] == -1
                            send[sd[r, a] + i] = (rnd << 32 | r, a << 32 | i, 0)
                    else:
                        assert sd[r, a] == -1
            want_send = [t3 for d in range(n) for t3 in H.exchange_block(owner, rank, d, g)]
            assert send[:sum(sc), :2].tolist() == [[rnd << 32 | r, a << 32 | i] for (r, a, i) in want_send], "round %d: send buffer out of order" % rnd
            recv = np.zeros((max(1, sum(rc)), H.REC_WORDS), np.int64)
            if int(g[:, 1:].sum()):                         # the same number on every rank: nobody enters an empty collective alone
                E = H.REC_WORDS
                st, rt = torch.from_numpy(send.reshape(-1)), torch.from_numpy(recv.reshape(-1))
                dist.all_to_all_single(rt[:sum(rc) * E], st[:sum(sc) * E], [c * E for c in rc], [c * E for c in sc])
            assert int(tot.value) == sum(rc)
            nxt = {}
            for (src, off, vsrc, dst, cnt) in pcs[:k].tolist():   # piece: cnt records of source rank src, at vsrc.. of the receive order, go to dst.. of the next arrays
                blk = H.exchange_block(owner, src, rank, g)
                base = sum(H.exchange_layout(owner, n, src, g)[:rank])
                for i in range(cnt):
                    r_, a_, i_ = blk[off - base + i]
                    assert recv[vsrc + i, :2].tolist() == [rnd << 32 | r_, a_ << 32 | i_], "round %d: record out of place" % rnd
                    nxt[dst + i] = (r_, a_, i_)
            want = []                                           # next arrays: this rank's pieces ascending; inside (a,b): sources (b,x) by x, old order
            for r2 in range(1, NR):
                if owner[r2] == rank:
                    a, b = H.rope_sym(r2), H.rope_prev(r2)
                    for r in range(NR):
                        if H.rope_sym(r) == b:
                            want += [(r, a, i) for i in range(int(g[r, a]))]
            assert [nxt[i] for i in range(len(want))] == want and len(nxt) == len(want)
        print("rank %d/%d plan exchange ok" % (rank, n))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
