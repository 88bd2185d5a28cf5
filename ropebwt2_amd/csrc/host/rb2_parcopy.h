/* rb2_parcopy.h -- a large memcpy by four threads.  One thread moves ~6 GB/s out of a pinned staging buffer; the run bytes of a big
 * index (7 GB at configs[1], 88 GB at configs[2]) are copied once on their way to the .fmd coder / the tree loader, and that one
 * copy was the longest single-threaded stretch of both paths. */
#ifndef RB2_PARCOPY_H_
#define RB2_PARCOPY_H_
#include <pthread.h>
#include <string.h>
#include <stdint.h>

typedef struct { uint8_t *d; const uint8_t *s; size_t n; } rb2_cpjob_t;
static void *rb2_cp_worker(void *a) { rb2_cpjob_t *j = (rb2_cpjob_t*)a; memcpy(j->d, j->s, j->n); return 0; }
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
#endif
