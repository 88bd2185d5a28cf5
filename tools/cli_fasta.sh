#!/bin/bash
# Run on the GPU box: long reads as FASTA (80 columns) through the CLI, file in /dev/shm: the threaded FASTA reader against the
# sequential kseq-exact one (RB2_SEQ_FASTX=1).  Default: 1 M x 10 kbp (configs[3] at one tenth); the .fmd md5 of both must agree.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1000000}
L=${2:-10000}
F=/dev/shm/rb2_long.fa
$R/ropebwt2_amd/bin/synth_reads $N $L 44 0 0 2 > $F
ls -la $F
for mode in threaded sequential; do
	echo "== $mode: ropebwt2 -d -m10g -o /dev/shm/long.fmd $F"
	if [ $mode = sequential ]; then export RB2_SEQ_FASTX=1; else unset RB2_SEQ_FASTX; fi
	( time RB2_PARSE_TRACE=1 $R/ropebwt2_amd/bin/ropebwt2 -d -m10g -v3 -o /dev/shm/long.fmd $F ) 2>&1 | grep -E "inserted|constructed|Real|real|parsed|FASTA|streamed"
	md5sum /dev/shm/long.fmd | cut -c1-32
	rm -f /dev/shm/long.fmd
done
rm -f $F
