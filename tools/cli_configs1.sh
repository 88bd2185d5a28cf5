#!/bin/bash
# Run on the GPU box: BASELINE.json configs[1] (100 M x 101 bp, one read per line) through the CLI into a .fmd file in /dev/shm:
# the index written while it is encoded (default) against the one-shot writer (RB2_FMD_NO_STREAM=1); md5 = configs[1]'s golden.
# $2: worker counts of the .fmd coder to try (RB2_FMD_THREADS; default: the CLI's own choice)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-100000000}
F=/dev/shm/rb2_c1.txt
$R/ropebwt2_amd/bin/synth_reads $N 101 42 > $F
ls -la $F
nproc
export RB2_FMD_STATS=1
for mode in streamed streamed oneshot ${2:-}; do
	echo "== $mode: ropebwt2 -LRds -m4g -v3 -o /dev/shm/c1.fmd $F"
	unset RB2_FMD_NO_STREAM RB2_FMD_THREADS
	case $mode in oneshot) export RB2_FMD_NO_STREAM=1;; streamed) ;; *) export RB2_FMD_THREADS=$mode;; esac
	( time $R/ropebwt2_amd/bin/ropebwt2 -LRds -m4g -v3 -o /dev/shm/c1.fmd $F ) 2>&1 | grep -E "inserted|constructed|Real|real|streamed|finish|rb2_fmdp"
	md5sum /dev/shm/c1.fmd | cut -c1-32
	rm -f /dev/shm/c1.fmd
done
rm -f $F
