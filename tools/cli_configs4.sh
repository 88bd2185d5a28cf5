#!/bin/bash
# Run on the GPU box: BASELINE.json configs[4] (`-bi old.fmr`: 500 M reads on top of an index of 500 M) at full size through the CLI,
# both phases, files in /dev/shm, input piped from the generator.  The md5 of the incremental .fmr must be round 2's
# (ca371349d6a84f1a4a996f1a0050a198 = the one-shot build of all 10^9 reads, profiles/r02_configs4_full_cli.txt).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-500000000}
grep -E "MemTotal|MemAvailable" /proc/meminfo
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
if [ "$avail" -lt 500 ] && [ "$N" -ge 500000000 ]; then echo "less than 500 GB available: not running the full size"; exit 0; fi
echo "== phase 1: synth_reads $N 101 42 | ropebwt2 -LRbs -o /dev/shm/old.fmr -"
( time ( $R/ropebwt2_amd/bin/synth_reads $N 101 42 | RB2_SYNC_TRACE=1 $R/ropebwt2_amd/bin/ropebwt2 -LRbs -v3 -o /dev/shm/old.fmr - ) ) 2>&1 | grep -E "constructed|written as|Real|real|symbol counts"
ls -la /dev/shm/old.fmr
echo "== phase 2: synth_reads $N 101 42 $N | ropebwt2 -LRbs -i /dev/shm/old.fmr -o /dev/shm/inc.fmr -"
( time ( $R/ropebwt2_amd/bin/synth_reads $N 101 42 $N | RB2_SYNC_TRACE=1 $R/ropebwt2_amd/bin/ropebwt2 -LRbs -v3 -i /dev/shm/old.fmr -o /dev/shm/inc.fmr - ) ) 2>&1 | grep -E "inserted|constructed|written as|Real|real|symbol counts|mr_restore"
ls -la /dev/shm/inc.fmr
md5sum /dev/shm/inc.fmr | cut -c1-32
rm -f /dev/shm/old.fmr /dev/shm/inc.fmr
