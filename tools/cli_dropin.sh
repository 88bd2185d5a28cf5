#!/bin/bash
# Run on the GPU box: the REFERENCE's own main.c linked against our library (oracle/_ref/ropebwt2_dropin, oracle/Makefile: dropin)
# on BASELINE.json configs[1] -- text in /dev/shm, .fmd out -- beside our CLI on the same file.  The reference's main() reads with
# kseq on one thread, calls mr_insert_multi per batch, and writes the .fmd by walking mr_itr_next_block on one thread (main.c:288-320).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-100000000}
F=/dev/shm/rb2_c1.txt
$R/ropebwt2_amd/bin/synth_reads $N 101 42 > $F
make -s -C $R/oracle dropin 2>/dev/null
for exe in $R/oracle/_ref/ropebwt2_dropin $R/ropebwt2_amd/bin/ropebwt2; do
	echo "== $exe -LRds -m4g -o /dev/shm/c1.fmd $F"
	( time RB2_SYNC_TRACE=1 $exe -LRds -m4g -o /dev/shm/c1.fmd $F ) 2>&1 | grep -E "inserted|constructed|Real|real|CPU|mr_sync|mr_itr|rope" | cut -c1-200
	md5sum /dev/shm/c1.fmd | cut -c1-32
	rm -f /dev/shm/c1.fmd
done
rm -f $F
