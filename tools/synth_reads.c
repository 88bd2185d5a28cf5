/* tools/synth_reads.c -- input generator for tests and bench.py (built to ropebwt2_amd/bin/synth_reads).
 * Deterministic synthetic read generator: base j of read i is
 * "ACGT"[splitmix64_output(seed, i*L+j+1) >> 62], one read per line.
 *   usage: synth_reads <n_reads> <read_len> [seed=42] [first_read=0] [genome_len=0] [fastq=0] [skew=0]  > reads.txt
 * skew = 1: skewed composition -- the top byte u of the same splitmix output picks the base: A if u < 218 (85 %), else "CGT"[(u - 218) % 3]
 * (5 % each): piece (A,A) of the index then holds 72 % of all symbols and passes 2^32 symbols at 59 M x 101 bp (the > 32-bit regime
 * of the position arithmetic at a size the reference builds in minutes; the same stream as rb2_hip_synth_reads_skew).
 * fastq = 1: the same reads as four-line FASTQ records ("@r<i>", bases, "+", a constant quality string of 'I')
 * fastq = 2: as FASTA records (">r<i>", bases wrapped at 80 columns)
 * genome_len > 0: read i is the window of a random genome (base p = "ACGT"[sm64(seed, p) >> 62]) that starts at
 * sm64(seed ^ COV_SALT, i) % (genome_len - L + 1) -- overlapping reads, the same stream as rb2_hip_synth_reads_cov.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

static inline uint64_t sm64(uint64_t seed, uint64_t k)
{
	uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s n_reads read_len [seed] [first_read]\n", argv[0]); return 1; }
	uint64_t n = strtoull(argv[1], 0, 10), L = strtoull(argv[2], 0, 10);
	uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 10) : 42;
	uint64_t first = argc > 4 ? strtoull(argv[4], 0, 10) : 0;
	uint64_t glen = argc > 5 ? strtoull(argv[5], 0, 10) : 0;
	const int fastq = argc > 6 ? atoi(argv[6]) : 0;
	const int skew = argc > 7 ? atoi(argv[7]) : 0;
	const uint64_t COV_SALT = 0x5bd1e995c0f3a1d7ULL;
	if (glen && glen < L) { fprintf(stderr, "genome shorter than a read\n"); return 1; }
	char *line = (char*)malloc(L + 2), *qual = (char*)malloc(L + 4);
	for (uint64_t j = 0; j < L; ++j) qual[j] = 'I';
	qual[L] = '\n';
	static char obuf[1 << 20];
	setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
	for (uint64_t i = first; i < first + n; ++i) {
		const uint64_t base = glen ? sm64(seed ^ COV_SALT, i) % (glen - L + 1) : i * L;
		if (!skew) for (uint64_t j = 0; j < L; ++j) line[j] = "ACGT"[sm64(seed, base + j) >> 62];
		else for (uint64_t j = 0; j < L; ++j) { const unsigned u = (unsigned)(sm64(seed, base + j) >> 56); line[j] = u < 218 ? 'A' : "CGT"[(u - 218) % 3]; }
		line[L] = '\n';
		if (fastq == 2) {
			printf(">r%llu\n", (unsigned long long)i);
			for (uint64_t j = 0; j < L; j += 80) { fwrite(line + j, 1, L - j < 80 ? L - j : 80, stdout); fputc('\n', stdout); }
			continue;
		}
		if (fastq) printf("@r%llu\n", (unsigned long long)i);
		fwrite(line, 1, L + 1, stdout);
		if (fastq) { fwrite("+\n", 1, 2, stdout); fwrite(qual, 1, L + 1, stdout); }
	}
	free(line); free(qual);
	return 0;
}
