#!/bin/bash
# Run on the GPU box: where the wall-clock of the configs[1] CLI run goes (-v4 phase lines), for the values of RB2_THP given
# (bit 1: batch buffers, 2: the .fmd array, 4: run-byte buffers get MADV_HUGEPAGE; default 7)
R=${GRAFT_REPO_ROOT:-/root/repo}
F=/dev/shm/rb2_c1.txt
$R/ropebwt2_amd/bin/synth_reads 100000000 101 42 > $F
for thp in ${@:-7}; do
echo "== RB2_THP=$thp"
( time RB2_THP=$thp $R/ropebwt2_amd/bin/ropebwt2 ${FLAGS:--LRds} -m4g -v4 -o /dev/shm/c1.fmd $F ) 2>&1 | grep -E "written as|mr_dump|set up|constructed|streamed|released|Real|real|written in"
rm -f /dev/shm/c1.fmd
done
rm -f $F
