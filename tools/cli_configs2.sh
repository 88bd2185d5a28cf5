#!/bin/bash
# Run on the GPU box: BASELINE.json configs[2] (1.2 B x 101 bp, -brR) THROUGH THE CLI on one GPU: the per-batch insert times the CLI
# prints (PCIe included; the batch being assembled is uploaded while the previous one is inserted), with the reference's default
# -m10g and with -m auto.  Output: the stderr lines of both runs.  The .fmr goes to /dev/shm and is removed.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1200000000}
for M in ${2:-10g auto}; do
	echo "== synth_reads $N 101 42 | ropebwt2 -LRbr -m$M -o /dev/shm/c2.fmr -"
	( time ( $R/ropebwt2_amd/bin/synth_reads $N 101 42 | RB2_SYNC_TRACE=1 $R/ropebwt2_amd/bin/ropebwt2 -LRbr -m$M -o /dev/shm/c2.fmr - ) ) 2>&1 | grep -E "inserted|constructed|auto|written as|mr_dump|Real|real|symbol counts"
	md5sum /dev/shm/c2.fmr | cut -c1-32
	rm -f /dev/shm/c2.fmr
done
