/* how fast can one file in /dev/shm be written: T threads of pwrite (8 MiB pieces) against T threads of memcpy into a shared mapping.
 * usage: tmpfs_write <GiB> <threads> <0 pwrite | 1 mmap>        (tools/ubench: the numbers behind the .fmr writer) */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static int fd, T, mode; static size_t N; static char *src, *map;
static void *work(void *a)
{
	const long k = (long)a;
	const size_t from = N / T * k, to = k == T - 1 ? N : N / T * (k + 1), step = 8 << 20;
	size_t o;
	for (o = from; o < to; o += step) {
		const size_t n = to - o < step ? to - o : step;
		if (mode) memcpy(map + o, src + (o & ((64 << 20) - 1) & ~(step - 1)), n);
		else if (pwrite(fd, src + (o & ((64 << 20) - 1) & ~(step - 1)), n, (off_t)o) != (ssize_t)n) { perror("pwrite"); exit(1); }
	}
	return 0;
}
int main(int argc, char **argv)
{
	pthread_t th[64]; long k; double t0, t1;
	N = (size_t)atol(argv[1]) << 30; T = atoi(argv[2]); mode = atoi(argv[3]);
	src = malloc(64 << 20); memset(src, 7, 64 << 20);
	fd = open("/dev/shm/rb2_tmpfs_write.bin", O_RDWR | O_CREAT | O_TRUNC, 0644);
	t0 = now();
	if (mode) { if (ftruncate(fd, (off_t)N)) { perror("ftruncate"); return 1; } map = mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
	for (k = 0; k < T; ++k) pthread_create(&th[k], 0, work, (void*)k);
	for (k = 0; k < T; ++k) pthread_join(th[k], 0);
	if (mode) munmap(map, N);
	t1 = now();
	printf("%s, %d threads: %.2f GB/s (%.3f s for %zu GiB)\n", mode ? "mmap + memcpy" : "pwrite", T, N / 1e9 / (t1 - t0), t1 - t0, N >> 30);
	close(fd); unlink("/dev/shm/rb2_tmpfs_write.bin");
	return 0;
}
