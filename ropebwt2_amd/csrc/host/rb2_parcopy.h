/* rb2_parcopy.h -- a large memcpy by four threads.  One thread moves ~6 GB/s out of a pinned staging buffer; the run bytes of a big
 * index (7 GB at configs[1], 88 GB at configs[2]) are copied once on their way to the .fmd coder / the tree loader, and that one
 * copy was the longest single-threaded stretch of both paths. */
#ifndef RB2_PARCOPY_H_
#define RB2_PARCOPY_H_
#include <pthread.h>
#include <string.h>
#include <stdint.h>

typedef struct { uint8_t *d; const uint8_t *s; size_t n; } rb2_cpjob_t;
static __attribute__((unused)) void *rb2_cp_worker(void *a) { rb2_cpjob_t *j = (rb2_cpjob_t*)a; memcpy(j->d, j->s, j->n); return 0; }
static inline void rb2_par_memcpy(uint8_t *d, const uint8_t *s, int64_t n)
{
	enum { T = 4 };
	rb2_cpjob_t job[T]; pthread_t th[T]; int k, started = 0;
	const int64_t part = (n / T + 63) & ~(int64_t)63;
	if (n < (8 << 20)) { memcpy(d, s, (size_t)n); return; }
	for (k = 0; k < T; ++k) {
		const int64_t o = k * part, len = o >= n ? 0 : (n - o < part || k == T - 1 ? n - o : part);
		job[k].d = d + o; job[k].s = s + o; job[k].n = (size_t)len;
		if (k > 0 && len > 0) { if (pthread_create(&th[k], 0, rb2_cp_worker, &job[k]) == 0) started |= 1 << k; else rb2_cp_worker(&job[k]); }
	}
	rb2_cp_worker(&job[0]);
	for (k = 1; k < T; ++k) if (started >> k & 1) pthread_join(th[k], 0);
}
/* n bytes of a regular file from offset off, by four threads (pread): one thread copies 3-6 GB/s out of the page cache, and the reader
 * of the FASTA / FASTQ blocks is one thread.  Returns the bytes read (short only at the end of the file). */
#include <unistd.h>
typedef struct { int fd; uint8_t *d; int64_t n, off, got; } rb2_prjob_t;
static __attribute__((unused)) void *rb2_pr_worker(void *a)
{
	rb2_prjob_t *j = (rb2_prjob_t*)a;
	while (j->got < j->n) {
		const ssize_t r = pread(j->fd, j->d + j->got, (size_t)(j->n - j->got), (off_t)(j->off + j->got));
		if (r <= 0) break;
		j->got += r;
	}
	return 0;
}
static inline int64_t rb2_par_pread(int fd, uint8_t *d, int64_t n, int64_t off)
{
	enum { T = 4 };
	rb2_prjob_t job[T]; pthread_t th[T]; int k, started = 0;
	const int64_t part = (n / T + 4095) & ~(int64_t)4095;
	int64_t got = 0;
	for (k = 0; k < T; ++k) {
		const int64_t o = k * part, len = o >= n ? 0 : (n - o < part || k == T - 1 ? n - o : part);
		job[k].fd = fd; job[k].d = d + o; job[k].n = len; job[k].off = off + o; job[k].got = 0;
		if (k > 0 && len > 0 && n >= (4 << 20)) { if (pthread_create(&th[k], 0, rb2_pr_worker, &job[k]) == 0) started |= 1 << k; else rb2_pr_worker(&job[k]); }
		else if (k > 0 && len > 0) rb2_pr_worker(&job[k]);
		if (k == 0) continue;
	}
	rb2_pr_worker(&job[0]);
	for (k = 1; k < T; ++k) if (started >> k & 1) pthread_join(th[k], 0);
	for (k = 0; k < T; ++k) { got += job[k].got; if (job[k].got < job[k].n) break; }   /* (the file ended inside part k) */
	return got;
}

/* A large fresh buffer is faulted in 4 KiB at a time: 0.14 s per GB on the MI355X hosts, and as much again to give it back -- the
 * 6 GB output array of the .fmd coder, the batch buffers and the run-byte buffers of a configs[1] run are 30 GB of that.  With
 * transparent huge pages (the hosts run THP in `madvise` mode) the same costs 0.04 s per GB (tools/ubench/thp_probe.c).  Call it
 * on a buffer of tens of MB or more right after (re)allocating it; a no-op where THP is off. */
#include <sys/mman.h>
#include <stdlib.h>
static inline void rb2_hint_huge_which(void *p, size_t n, int which)
{
#ifdef MADV_HUGEPAGE
	static int mask = -1;
	if (mask < 0) { const char *e = getenv("RB2_THP"); mask = e ? atoi(e) : 15; }   /* 1: batch buffers, 4: run-byte buffers, 8: the .fmd array when its size is known up front */
	if (!(mask & which)) return;
	const size_t H = (size_t)2 << 20;
	const size_t a = ((size_t)p + H - 1) & ~(H - 1), e = ((size_t)p + n) & ~(H - 1);
	if (p && n >= ((size_t)32 << 20) && e > a) (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
#else
	(void)p; (void)n;
#endif
}
#define rb2_hint_huge(p, n) rb2_hint_huge_which(p, n, RB2_THP_WHICH)
#endif
