#!/bin/bash
# Run on the GPU box (needs ~350 GB of RAM): BASELINE.json configs[2] (1.2 B x 101 bp, -brR) through the CLI with the reads in a FILE in
# /dev/shm instead of a pipe from the generator (which bounds tools/cli_configs2.sh at ~1 GB/s of text): what the program itself does.
# The text is written by 8 generators side by side (not timed).  Output: the stderr lines and the .fmr md5 (165076bff6dc4a162026d0e2b97c1e0a).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1200000000}
M=${2:-auto}
L=${3:-101}                  # read length (configs[3]: 10000 with N = 10000000, FLAGS = -LRb, M = 10g, SEED = 44)
FLAGS=${4:--LRbr}
SEED=${5:-42}
F=/dev/shm/rb2_c2.txt
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
df -h /dev/shm | tail -1
if [ "$avail" -lt 600 ] && [ "$((N * L))" -ge 100000000000 ]; then echo "less than 600 GB available: not running the full size"; exit 0; fi
G=8; per=$((N / G)); t0=$(date +%s)
rm -f $F; truncate -s $((N * (L + 1))) $F
for k in $(seq 0 $((G - 1))); do
	n=$per; [ $k = $((G - 1)) ] && n=$((N - per * (G - 1)))
	( $R/ropebwt2_amd/bin/synth_reads $n $L $SEED $((k * per)) | dd of=$F bs=16M iflag=fullblock oflag=seek_bytes seek=$((k * per * (L + 1))) conv=notrunc status=none ) &
done
wait
echo "text written in $(( $(date +%s) - t0 )) s"; ls -la $F
for m in $M; do
	echo "== ropebwt2 $FLAGS -m$m -v4 -o /dev/shm/c2.fmr $F"
	( time RB2_SYNC_TRACE=1 $R/ropebwt2_amd/bin/ropebwt2 $FLAGS -m$m -v4 -o /dev/shm/c2.fmr $F ) 2>&1 | grep -E "batch done|constructed|auto|written as|released|set up|Real|real|symbol counts"
	ls -la /dev/shm/c2.fmr
	md5sum /dev/shm/c2.fmr | cut -c1-32
	rm -f /dev/shm/c2.fmr
done
rm -f $F
