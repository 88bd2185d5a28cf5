#!/usr/bin/env python3
"""Large-configuration check on ONE GPU: build the BWT of `--reads` synthetic reads batch by batch (each batch is
generated on the device just before it is inserted; only the inserts are timed) and verify size-independent
properties of the result: the count matrix sums, LF-consistency (#a in rope b == #rows of rope a that are followed
by b, summed over sub-ropes -- here: column sums equal rope sizes), sampled rank queries.

    python tools/scale_check.py --reads 1200000000 --batch 10 --order rclo      # BASELINE.json configs[2] on one GPU
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400_000_000)
    ap.add_argument("--read-len", type=int, default=101)
    ap.add_argument("--batch", type=float, default=10.0, help="-m in GiB (ropebwt2 default: 10g)")
    ap.add_argument("--order", default="rlo", choices=["io", "rlo", "rclo"])
    ap.add_argument("--both-strands", action="store_true")
    ap.add_argument("--genome-len", type=int, default=0, help="> 0: reads are windows of one random genome of this many bases (coverage data)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--profile", action="store_true", help="per-kernel times (hipEvents) and leaf-layout statistics in the output")
    args = ap.parse_args()
    from ropebwt2_amd import HipBwt, build_all
    build_all()
    L = args.read_len
    per_read = (L + 1) * (2 if args.both_strands else 1)
    per_batch = -(-(int(args.batch * 1024 ** 3 * 0.97) + 1) // per_read)       # main.c:136, 238
    so = {"io": 0, "rlo": 1, "rclo": 2}[args.order]
    bwt = HipBwt(so, 0)
    if args.profile:
        bwt.profile(True)
    total = args.reads * per_read
    bwt.reserve(per_batch * per_read, per_batch * (2 if args.both_strands else 1), total)
    buf = bwt.dev_alloc(per_batch * per_read + 64)
    done, times, per_batch_ms, prev_ms = 0, [], [], {}
    while done < args.reads:
        n = min(per_batch, args.reads - done)
        bwt.synth_reads(buf, done, n, L, seed=args.seed, strand=1 if args.both_strands else 0, genome_len=args.genome_len)
        bwt.sync()
        t0 = time.perf_counter()
        bwt.insert_multi_dev(buf, n * per_read)
        bwt.sync()
        times.append(time.perf_counter() - t0)
        done += n
        if args.profile:                                          # kernel time of THIS batch (the counters are cumulative)
            now = {k: v["ms"] for k, v in bwt.profile_get().items()}
            per_batch_ms.append({k: round(v - prev_ms.get(k, 0.0), 1) for k, v in now.items() if v - prev_ms.get(k, 0.0) >= 0.05})
            prev_ms = now
        sys.stderr.write("[scale] %d / %d reads, batch %.2f s (%.2f Gsym/s)\n" % (done, args.reads, times[-1], n * per_read / times[-1] / 1e9))
    c = bwt.counts()
    n_str = args.reads * (2 if args.both_strands else 1)
    ok = int(c.sum()) == total and int(c[:, 0].sum()) == n_str
    # LF-consistency: the number of rows that start with a (= size of rope a) equals the number of a's in the BWT
    sizes, occ = c.sum(axis=1), c.sum(axis=0)
    ok_lf = all(int(sizes[a]) == int(occ[a]) for a in range(1, 6)) and int(sizes[0]) == n_str
    # rank spot checks: rank at the end of rope b equals the row of the matrix; rank is monotone
    ok_rank = True
    rng = np.random.RandomState(1)
    for b in range(1, 5):
        n_b = int(sizes[b])
        ok_rank &= bool(np.array_equal(bwt.rank1a(b, n_b), c[b]))
        xs = np.sort(rng.randint(0, n_b + 1, size=6))
        prev = np.zeros(6, np.int64)
        for x in xs:
            r = bwt.rank1a(b, int(x))
            ok_rank &= int(r.sum()) == int(x) and bool(np.all(r >= prev))
            prev = r
    try:
        lay = bwt.layout_stats()                      # (+ re-spreads and leaf splits; an older library in an A/B run has only the four)
    except Exception:  # noqa: BLE001
        lay = bwt.sparse_stats()
    extra = {"layout": lay, "sparse_lambda": os.environ.get("RB2_SPARSE_LAMBDA", "default"), "library": os.environ.get("RB2_HIP_LIB", "this build")}
    if args.profile:
        extra["kernels_ms"] = {k: round(v["ms"], 2) for k, v in bwt.profile_get().items()}
        extra["kernels_ms_per_batch"] = per_batch_ms
    bwt.dev_free(buf)
    bwt.close()
    print(json.dumps({**extra, "reads": args.reads, "read_len": L, "order": args.order, "both_strands": args.both_strands, "genome_len": args.genome_len, "batch_gib": args.batch,
                      "batches": len(times), "symbols": total, "insert_s": sum(times), "gsym_per_s": total / sum(times) / 1e9,
                      "batch_s": [round(t, 3) for t in times], "counts_ok": ok, "lf_ok": ok_lf, "rank_ok": ok_rank}))


if __name__ == "__main__":
    main()
