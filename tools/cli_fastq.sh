#!/bin/bash
# Run on the GPU box: BASELINE.json configs[1] as FOUR-LINE FASTQ (the format real inputs have) through the CLI, file in /dev/shm:
# the threaded FASTQ reader against the sequential kseq-exact one (RB2_SEQ_FASTX=1); the .fmd md5 must be configs[1]'s golden.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-100000000}
F=/dev/shm/rb2_c1.fq
$R/ropebwt2_amd/bin/synth_reads $N 101 42 0 0 1 > $F
ls -la $F
for mode in threaded sequential; do
	echo "== $mode: ropebwt2 -Rds -m4g -o /dev/shm/c1.fmd $F"
	if [ $mode = sequential ]; then export RB2_SEQ_FASTX=1; else unset RB2_SEQ_FASTX; fi
	( time $R/ropebwt2_amd/bin/ropebwt2 -Rds -m4g -o /dev/shm/c1.fmd $F ) 2>&1 | grep -E "inserted|constructed|Real|real|parsed|FASTQ"
	md5sum /dev/shm/c1.fmd | cut -c1-32
	rm -f /dev/shm/c1.fmd
done
rm -f $F
