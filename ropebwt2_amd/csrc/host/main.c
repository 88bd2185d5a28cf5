/* main.c -- command-line front end with the flags and outputs of `ropebwt2` (reference:
 * /root/reference/main.c:89-343), on top of include/mrope.h.  Batches are inserted by the GPU
 * engine; `-m0` (one string at a time, mr_insert1) is the only mode that builds on the CPU.
 *
 * Own code: option handling, a small FASTA/FASTQ/line reader over zlib, read filters, the three
 * writers (plain text, FMR, FMD).  The undocumented CRLF output (-B) of the reference is not
 * provided.
 */
#define _GNU_SOURCE                  /* memrchr */
#include <zlib.h>
#include <fcntl.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <pthread.h>
#include "mrope.h"
#include "rle.h"
#include "rb2_fmd.h"
#define RB2_THP_WHICH 1
#include "rb2_parcopy.h"

#define RB2_VERSION "r187-hip1"

/* ---- tiny growable byte string ---------------------------------------------------------------- */
typedef struct { size_t l, m; char *s; } str_t;
/* the CLI cannot go on without its buffers: say so instead of writing through a null pointer */
static void *xrealloc(void *p, size_t n)
{
	void *q = realloc(p, n ? n : 1);
	if (q == 0) { fprintf(stderr, "[E::%s] out of memory (%zu bytes)\n", "main_ropebwt2", n); exit(1); }
	return q;
}
static void str_reserve(str_t *s, size_t need) { if (need > s->m) { s->m = need + (need >> 1) + 64; s->s = (char*)xrealloc(s->s, s->m); rb2_hint_huge(s->s, s->m); } }
static void str_putc(str_t *s, int c) { str_reserve(s, s->l + 2); s->s[s->l++] = (char)c; s->s[s->l] = 0; }
static void str_append(str_t *s, const char *p, size_t n) { str_reserve(s, s->l + n + 1); memcpy(s->s + s->l, p, n); s->l += n; s->s[s->l] = 0; }

/* ---- buffered reader: FASTA / FASTQ records or plain lines ------------------------------------
 * Behaviour follows kseq.h as the reference uses it (main.c:177-187), including what happens at the edges:
 * the stream is refilled RD_BUF bytes at a time and "end of input" is only known once a refill comes back short
 * (kseq.h:70-75, 96-105) -- so an input whose size is a multiple of RD_BUF (0 included) yields one extra empty line
 * in -L mode; sequence lines are taken raw (embedded blanks become N), a trailing CR is dropped; a FASTQ record
 * whose quality string is shorter than its sequence ends the input (kseq_read's -2). */
#define RD_BUF 16384
typedef struct {
	gzFile fp; unsigned char buf[RD_BUF]; int beg, end, eof, last; str_t seq, qual;
	/* raw blocks to be read BEFORE the stream goes on (the threaded FASTQ reader hands its unparsed blocks back when the input
	 * turns out not to be strict four-line FASTQ); pos = stream offset of the next refill: refills stay on multiples of RD_BUF,
	 * as in a run that read the stream sequentially from its start */
	const uint8_t **mem; const int64_t *mem_n; int nmem, imem; int64_t mem_off, pos;
} reader_t;

static int rd_fill(reader_t *r)               /* 0: nothing more to read */
{
	const int want = RD_BUF - (int)(r->pos % RD_BUF);
	int got = 0;
	r->beg = 0;
	while (got < want && r->imem < r->nmem) {
		int64_t k = r->mem_n[r->imem] - r->mem_off;
		if (k > want - got) k = want - got;
		memcpy(r->buf + got, r->mem[r->imem] + r->mem_off, (size_t)k);
		got += (int)k; r->mem_off += k;
		if (r->mem_off == r->mem_n[r->imem]) { ++r->imem; r->mem_off = 0; }
	}
	if (got < want) { const int k = gzread(r->fp, r->buf + got, (unsigned)(want - got)); if (k > 0) got += k; }
	r->end = got; r->pos += got;
	if (r->end < want) r->eof = 1;
	return r->end > 0;
}
static int rd_getc(reader_t *r)
{
	if (r->beg >= r->end) {
		if (r->eof || !rd_fill(r)) return -1;
	}
	return r->buf[r->beg++];
}
/* bytes up to the next delimiter ('\n', or any white space when !line) are appended to s (NULL: skipped); *delim gets
 * the delimiter found (0: input ended first).  Returns -1 only when the input was already known to be exhausted. */
static int rd_until(reader_t *r, str_t *s, int line, int *delim)
{
	if (delim) *delim = 0;
	if (r->beg >= r->end && r->eof) return -1;
	for (;;) {
		unsigned char *b;
		int i, n;
		if (r->beg >= r->end) {
			if (r->eof || !rd_fill(r)) break;
		}
		b = r->buf + r->beg; n = r->end - r->beg;
		if (line) { unsigned char *nl = (unsigned char*)memchr(b, '\n', n); i = nl ? (int)(nl - b) : n; }
		else for (i = 0; i < n && !isspace(b[i]); ++i);
		if (s && i) str_append(s, (const char*)b, i);
		r->beg += i + 1;
		if (i < n) { if (delim) *delim = b[i]; break; }
	}
	if (s) {
		str_reserve(s, s->l + 1);
		if (line && s->l > 1 && s->s[s->l-1] == '\r') --s->l;      /* kseq.h:136 */
		s->s[s->l] = 0;
	}
	return 0;
}
static int read_line_record(reader_t *r)
{
	r->seq.l = 0; r->qual.l = 0;
	if (rd_until(r, &r->seq, 1, 0) < 0) return -1;
	return (int)r->seq.l;
}
static int read_fastx_record(reader_t *r)
{
	int c, d;
	if (r->last == 0) {                                  /* look for the next header */
		while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@');
		if (c < 0) return -1;
		r->last = c;
	}
	r->seq.l = r->qual.l = 0;
	str_reserve(&r->seq, 1); r->seq.s[0] = 0;
	if (rd_until(r, 0, 0, &d) < 0) return -1;            /* name; then the comment up to the end of the line */
	if (d != '\n') rd_until(r, 0, 1, 0);
	while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@' && c != '+') {
		if (c == '\n') continue;
		str_putc(&r->seq, c);
		rd_until(r, &r->seq, 1, 0);
	}
	if (c == '>' || c == '@') r->last = c;
	if (c != '+') return (int)r->seq.l;
	while ((c = rd_getc(r)) >= 0 && c != '\n');          /* rest of the '+' line */
	if (c < 0) return -2;
	str_reserve(&r->qual, 1); r->qual.s[0] = 0;
	while (rd_until(r, &r->qual, 1, 0) >= 0 && r->qual.l < r->seq.l);
	r->last = 0;
	if (r->qual.l != r->seq.l) return -2;                /* truncated quality: the reference stops reading here */
	return (int)r->seq.l;
}

/* ---- helpers ---------------------------------------------------------------------------------- */
/* output: one chunk of 43+3 run bytes -> .fmd encoder (user != 0) or plain text */
static void emit_runs(void *user, const uint8_t *q, int64_t n)
{
	const uint8_t *end = q + n;
	if (user) { rb2_fmdp_push_runs((rb2_fmdp_t*)user, q, n); return; }
	while (q < end) {
		int sym; int64_t len, k;
		rle_dec1(q, sym, len);
		{                                                      /* plain text: one fwrite per 64 KiB instead of a putchar per symbol */
			static char tbuf[1 << 16];
			static size_t tl = 0;
			for (k = len; k > 0; ) {
				const size_t room = sizeof(tbuf) - tl, take = (size_t)k < room ? (size_t)k : room;
				memset(tbuf + tl, "$ACGTN"[sym], take);
				tl += take; k -= (int64_t)take;
				if (tl == sizeof(tbuf)) { fwrite(tbuf, 1, tl, stdout); tl = 0; }
			}
			if (q == end && tl) { fwrite(tbuf, 1, tl, stdout); tl = 0; }   /* end of chunk: keep stdout in order with the caller's putchar */
		}
	}
}

static double cputime(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec); }
static double realtime(void) { struct timeval t; gettimeofday(&t, 0); return t.tv_sec + 1e-6 * t.tv_usec; }

static uint8_t nt6_tab[256];                            /* main.c:17-26: $ACGTN = 0..5, everything else N */
static void nt6_init(void)
{
	int i;
	for (i = 0; i < 256; ++i) nt6_tab[i] = 5;
	nt6_tab[0] = 0;
	nt6_tab['A'] = nt6_tab['a'] = 1; nt6_tab['C'] = nt6_tab['c'] = 2;
	nt6_tab['G'] = nt6_tab['g'] = 3; nt6_tab['T'] = nt6_tab['t'] = 4;
}
static int comp6(int c) { return c >= 1 && c <= 4 ? 5 - c : c; }
static int is_own_revcomp(int l, const uint8_t *s)      /* even length and s == revcomp(s) (main.c:80-87) */
{
	int i;
	if (l & 1) return 0;
	for (i = 0; i < l / 2; ++i) if (s[i] + s[l-1-i] != 5) return 0;
	return 1;
}

enum { F_FOR = 1, F_REV = 2, F_ODD = 4, F_BIN = 8, F_TREE = 16, F_THR = 64, F_LINE = 256, F_RLD = 512, F_NON = 1024, F_CUTN = 4096 };

/* ---- from a raw record to the strings handed to the index (main.c:184-237) --------------------- */
typedef struct { int flag, min_q, min_cut, batch; } enc_cfg_t;

/* everything between the reader and the strands: -L trimming, nt6 codes, quality masking, -N, reversal, -x / -C.  Works in
 * place on s[0..l] (s[l] is writable).  Returns the new length, or -1 when the record is dropped. */
static int prepare_record(const enc_cfg_t *c, uint8_t *s, int l, const char *qual, int qual_l)
{
	const int flag = c->flag;
	int i;
	if (flag & F_LINE) { for (i = 0; i < l && (((s[i] | 32) - 'a') < 26u); ++i); l = i; }   /* keep the leading letters (main.c:184-187) */
	for (i = 0; i < l; ++i) s[i] = nt6_tab[s[i]];
	if (!(flag & F_LINE) && qual_l && c->min_q > 0)
		for (i = 0; i < l && i < qual_l; ++i) if (qual[i] - 33 < c->min_q) s[i] = 5;
	if (flag & F_NON) { for (i = 0; i < l && s[i] != 5; ++i); if (i < l) return -1; }
	for (i = 0; i < l / 2; ++i) { uint8_t t = s[i]; s[i] = s[l-1-i]; s[l-1-i] = t; }   /* the API wants reversed strings */
	s[l] = 0;
	if (flag & F_CUTN) {                                 /* split at N, drop short pieces (main.c:204-218) */
		int k = 0, b = 0;
		for (i = 0; i <= l; ++i) {
			if (i == l || s[i] == 5) {
				const int seg = i - b;
				if (seg >= c->min_cut) {
					if ((flag & F_ODD) && is_own_revcomp(seg, &s[k - seg])) --k;
					s[k++] = 0;
				} else k -= seg;
				b = i + 1;
			} else s[k++] = s[i];
		}
		if (--k <= 0) return -1;
		l = k;
	} else if ((flag & F_ODD) && is_own_revcomp(l, s)) {
		if (l > 0) s[--l] = 0;
		else if (c->batch) return -1;                    /* reference quirk (main.c:219-222): an empty read under -C vanishes in batch mode */
	}
	return l;
}

static void revcomp_in_place(uint8_t *s, int l)         /* reverse complement, again stored reversed: complement in place */
{
	int i;
	for (i = 0; i < l / 2; ++i) { uint8_t t = (uint8_t)comp6(s[l-1-i]); s[l-1-i] = (uint8_t)comp6(s[i]); s[i] = t; }
	if (l & 1) s[l/2] = (uint8_t)comp6(s[l/2]);
}

static void append_strands(const enc_cfg_t *c, uint8_t *s, int l, str_t *out)
{
	if (c->flag & F_FOR) str_append(out, (char*)s, l + 1);
	if (c->flag & F_REV) { revcomp_in_place(s, l); str_append(out, (char*)s, l + 1); }
}

/* ---- -L input in batch mode: lines are independent, so blocks of whole lines are encoded by worker threads while the main
 * thread reads ahead and appends the finished blocks, in order, to the batch buffer (same strings, same batch boundaries as
 * the sequential loop; at GPU insert rates the text parser is the wall-clock of the whole program). ---- */
typedef struct {
	uint8_t *in; int64_t n_in, m_in; int extra_empty;    /* whole lines (the last one may lack its newline) + kseq's phantom empty line */
	str_t out; uint32_t *rec_end; size_t n_rec, m_rec;   /* encoded strings; out offset after every record */
	int state;                                           /* 0 free, 1 queued, 2 done */
	int at_eof, failed; int64_t stream_off;              /* FASTQ mode: last block of the input; not strict four-line FASTQ (in[] untouched); offset of in[0] in the stream */
} pjob_t;
typedef struct {
	enc_cfg_t cfg; pjob_t *job; int njob;
	int64_t next_work, n_queued; int closing;
	int64_t consumed; int eof;                                  /* blocks the main thread has taken over; the reader thread saw the end of the input */
	gzFile fp; int64_t chunk;                                   /* the reader thread's input */
	int fd;                                                     /* >= 0: the input is a plain (uncompressed) regular file -- read(2) it directly; zlib's transparent mode copies every byte twice */
	int fastq, stop;                                            /* 1: blocks of whole four-line FASTQ records, 2: of whole FASTA records, instead of whole lines; the consumer asks the reader to stop (fallback) */
	uint8_t *carry; int64_t n_carry, total;                     /* what the reader thread held back when it stopped, and how far it had read */
	int direct; int64_t fsize; int last_is_nl;                  /* -L on a plain regular file: no reader thread -- block k is the lines that START in bytes [k, k + 1) * chunk, the workers pread them */
	pthread_mutex_t mu; pthread_cond_t cv_work, cv_done, cv_space;
} pparse_t;

static void pjob_encode(const enc_cfg_t *cfg, pjob_t *jb)
{
	uint8_t *p = jb->in, *end = jb->in + jb->n_in;
	jb->out.l = 0; jb->n_rec = 0;
	str_reserve(&jb->out, (size_t)jb->n_in * (((cfg->flag & F_FOR) ? 1 : 0) + ((cfg->flag & F_REV) ? 1 : 0)) + 64);
	while (p < end || jb->extra_empty) {
		uint8_t *nl, *next;
		int l;
		if (p < end) {
			nl = (uint8_t*)memchr(p, '\n', end - p);
			next = nl ? nl + 1 : end;
			if (!nl) nl = end;
			l = (int)(nl - p);
			if (l > 1 && p[l-1] == '\r') --l;               /* kseq.h:136 */
		} else { jb->extra_empty = 0; next = end; l = 0; }    /* (in[n_in] is writable) */
		l = prepare_record(cfg, p, l, 0, 0);
		if (l >= 0) {
			append_strands(cfg, p, l, &jb->out);
			if (jb->n_rec == jb->m_rec) { jb->m_rec = jb->m_rec ? jb->m_rec * 2 : 1 << 16; jb->rec_end = (uint32_t*)xrealloc(jb->rec_end, jb->m_rec * 4); }
			jb->rec_end[jb->n_rec++] = (uint32_t)jb->out.l;
		}
		p = next;
	}
}

/* ---- FASTQ in batch mode, threaded.  Real inputs are FASTQ; kseq's grammar (multi-line sequences and qualities, '@' and '>' are
 * only special at the start of a line, a quality string is as long as its sequence says) is sequential in general -- but a file of
 * strict FOUR-LINE records is not: blocks cut after a multiple of four lines start at a record, and every record can be checked
 * on its own to be one that kseq reads exactly like four lines (header starts with '@', third line with '+', the sequence line
 * does not start with '+', '@' or '>', quality as long as the sequence).  Workers verify while they encode and never write to the
 * block; the first block that is NOT strict sends the consumer back to the sequential reader, which takes the raw blocks from
 * that one on and then the rest of the stream -- so the result is kseq's for every input, and four-line FASTQ is parsed by all cores. */
static int pjob_encode_fastq(const enc_cfg_t *cfg, pjob_t *jb, str_t *tmp)
{
	const uint8_t *p = jb->in, *end = jb->in + jb->n_in;
	jb->out.l = 0; jb->n_rec = 0;
	str_reserve(&jb->out, (size_t)jb->n_in / 2 * (((cfg->flag & F_FOR) ? 1 : 0) + ((cfg->flag & F_REV) ? 1 : 0)) + 64);
	while (p < end) {
		const uint8_t *ln[4]; int64_t len[4]; int i, sl, ql, l;
		for (i = 0; i < 4; ++i) {
			const uint8_t *nl;
			if (p >= end) return 0;                              /* fewer than four lines left */
			nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
			if (!nl) { if (!(jb->at_eof && i == 3)) return 0; nl = end; }   /* only the very last line of the input may lack its newline */
			ln[i] = p; len[i] = nl - p; p = nl < end ? nl + 1 : end;
		}
		if (len[0] < 1 || ln[0][0] != '@') return 0;
		if (len[2] < 1 || ln[2][0] != '+') return 0;
		if (len[1] > 0 && (ln[1][0] == '+' || ln[1][0] == '@' || ln[1][0] == '>')) return 0;
		if (len[1] > 0x3fffffff) return 0;
		sl = (int)len[1]; if (sl > 1 && ln[1][sl-1] == '\r') --sl;   /* kseq.h:136, as rd_until does */
		ql = (int)len[3]; if (ql > 1 && ln[3][ql-1] == '\r') --ql;
		if (ql != sl) return 0;                                /* shorter: kseq reads on; longer: kseq gives up -- the sequential reader does either */
		str_reserve(tmp, (size_t)sl + 2);
		memcpy(tmp->s, ln[1], (size_t)sl);
		l = prepare_record(cfg, (uint8_t*)tmp->s, sl, (const char*)ln[3], ql);
		if (l >= 0) {
			append_strands(cfg, (uint8_t*)tmp->s, l, &jb->out);
			if (jb->n_rec == jb->m_rec) { jb->m_rec = jb->m_rec ? jb->m_rec * 2 : 1 << 16; jb->rec_end = (uint32_t*)xrealloc(jb->rec_end, jb->m_rec * 4); }
			jb->rec_end[jb->n_rec++] = (uint32_t)jb->out.l;
		}
	}
	return 1;
}

/* ---- FASTA in batch mode, threaded (long reads and assemblies come as FASTA).  '>' is only special at the start of a line, and a
 * sequence line cannot start with it: a block cut in front of a line that starts with '>' starts at a record.  Workers read the
 * plain case -- header line, then sequence lines (concatenated, empty ones skipped: kseq.h:190-196) up to the next '>' line -- and
 * hand everything else back to the sequential kseq-exact reader like the FASTQ workers do: a line that starts with '+' or '@'
 * (kseq reads a quality string / a new record there), a carriage return (kseq strips it depending on what it has gathered so far),
 * a last line without its newline. */
static int pjob_encode_fasta(const enc_cfg_t *cfg, pjob_t *jb, str_t *tmp)
{
	const uint8_t *p = jb->in, *end = jb->in + jb->n_in;
	jb->out.l = 0; jb->n_rec = 0;
	str_reserve(&jb->out, (size_t)jb->n_in * (((cfg->flag & F_FOR) ? 1 : 0) + ((cfg->flag & F_REV) ? 1 : 0)) + 64);
	if (p < end && end[-1] != '\n') return 0;
	while (p < end) {
		const uint8_t *nl;
		int64_t sl = 0;
		int l;
		if (*p != '>') return 0;
		nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));   /* header: name and comment are not used */
		p = nl + 1;                                              /* (the block ends with a newline) */
		while (p < end && *p != '>') {
			int64_t len;
			nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
			len = nl - p;
			if (len > 0) {
				if (*p == '+' || *p == '@' || memchr(p, '\r', (size_t)len)) return 0;
				if (sl + len > 0x3fffffff) return 0;
				str_reserve(tmp, (size_t)(sl + len) + 2);
				memcpy(tmp->s + sl, p, (size_t)len); sl += len;
			}
			p = nl + 1;
		}
		str_reserve(tmp, (size_t)sl + 2);
		l = prepare_record(cfg, (uint8_t*)tmp->s, (int)sl, 0, 0);
		if (l >= 0) {
			const size_t before = jb->out.l;
			append_strands(cfg, (uint8_t*)tmp->s, l, &jb->out);
			if (jb->out.l > 0xffffffffu) { jb->out.l = before; return 0; }   /* rec_end is 32 bit: a block of > 4 GB of codes (one huge record, both strands) */
			if (jb->n_rec == jb->m_rec) { jb->m_rec = jb->m_rec ? jb->m_rec * 2 : 1 << 16; jb->rec_end = (uint32_t*)xrealloc(jb->rec_end, jb->m_rec * 4); }
			jb->rec_end[jb->n_rec++] = (uint32_t)jb->out.l;
		}
	}
	return 1;
}

/* -L on a plain regular file: the blocks need no reader thread to cut them -- one thread copying the file out of the page cache
 * (6 GB/s from tmpfs) was the whole "read + parse + insert" time of configs[1] once the parsing was spread over the cores.  Block k is
 * the lines that START in bytes [k * chunk, (k + 1) * chunk): its worker reads that range (and the byte in front of it, to know
 * whether a line starts at its first byte), drops the tail of the line that started earlier, and reads on to the end of its last line. */
static void pjob_fill_direct(pparse_t *pp, pjob_t *jb, int64_t k)
{
	const int64_t C = pp->chunk, size = pp->fsize, b0 = k * C, b1 = b0 + C < size ? b0 + C : size, from = b0 > 0 ? b0 - 1 : 0;
	int64_t have = 0, start, end;
	jb->stream_off = b0; jb->failed = 0; jb->n_in = 0; jb->at_eof = 0;
	jb->extra_empty = (k + 1) * C >= size && size % RD_BUF == 0 && pp->last_is_nl;   /* kseq's phantom empty line (see pparse_reader), in the last block */
	if (jb->m_in < 16) { jb->m_in = 16; jb->in = (uint8_t*)xrealloc(jb->in, jb->m_in); }   /* (pjob_encode may write in[n_in]) */
	if (b1 <= from) return;
	if (jb->m_in < (b1 - from) + 2) { jb->m_in = (b1 - from) + C + 2; jb->in = (uint8_t*)xrealloc(jb->in, jb->m_in); }
	while (have < b1 - from) {
		const ssize_t r = pread(pp->fd, jb->in + have, (size_t)(b1 - from - have), (off_t)(from + have));
		if (r <= 0) break;                                      /* the file shrank under us: what is there is the input */
		have += r;
	}
	if (b0 == 0) start = 0;
	else { const uint8_t *q = (const uint8_t*)memchr(jb->in, '\n', (size_t)have); start = q ? q - jb->in + 1 : have; }
	if (start >= have) return;                                 /* no line starts in this block */
	end = have;
	while (jb->in[end - 1] != '\n' && from + end < size) {      /* the last line goes on behind the block */
		const int64_t more = size - (from + end) < C ? size - (from + end) : C;
		const uint8_t *q;
		ssize_t r;
		if (jb->m_in < end + more + 2) { jb->m_in = end + more + C + 2; jb->in = (uint8_t*)xrealloc(jb->in, jb->m_in); }
		r = pread(pp->fd, jb->in + end, (size_t)more, (off_t)(from + end));
		if (r <= 0) break;
		q = (const uint8_t*)memchr(jb->in + end, '\n', (size_t)r);
		end = q ? q - jb->in + 1 : end + r;
	}
	memmove(jb->in, jb->in + start, (size_t)(end - start));
	jb->n_in = end - start;
	jb->at_eof = from + end >= size;
}

static void *pparse_worker(void *arg)
{
	pparse_t *pp = (pparse_t*)arg;
	pthread_mutex_lock(&pp->mu);
	for (;;) {
		pjob_t *jb;
		if (pp->direct) {                                       /* claim the next block as soon as its slot has been taken over by the consumer */
			int64_t k;
			while (pp->next_work < pp->n_queued && pp->next_work - pp->consumed >= pp->njob && !pp->closing) pthread_cond_wait(&pp->cv_space, &pp->mu);
			if (pp->next_work >= pp->n_queued || pp->closing) break;
			k = pp->next_work++;
			jb = &pp->job[k % pp->njob];
			pthread_mutex_unlock(&pp->mu);
			pjob_fill_direct(pp, jb, k);
			pjob_encode(&pp->cfg, jb);
			pthread_mutex_lock(&pp->mu);
			jb->state = 2;
			pthread_cond_broadcast(&pp->cv_done);
			continue;
		}
		while (pp->next_work >= pp->n_queued && !pp->closing) pthread_cond_wait(&pp->cv_work, &pp->mu);
		if (pp->next_work >= pp->n_queued) break;
		jb = &pp->job[pp->next_work++ % pp->njob];
		pthread_mutex_unlock(&pp->mu);
		if (pp->fastq) { str_t tmp = { 0, 0, 0 }; jb->failed = !(pp->fastq == 2 ? pjob_encode_fasta(&pp->cfg, jb, &tmp) : pjob_encode_fastq(&pp->cfg, jb, &tmp)); free(tmp.s); }
		else pjob_encode(&pp->cfg, jb);
		pthread_mutex_lock(&pp->mu);
		jb->state = 2;
		pthread_cond_broadcast(&pp->cv_done);
	}
	pthread_mutex_unlock(&pp->mu);
	return 0;
}

/* The reader of the parallel -L path, a thread of its own: reads the next block, cuts it behind its last newline (the unfinished
 * line is carried into the next block) and queues it for the encoders.  The main thread only takes the encoded blocks over, in
 * order -- with both in one thread, reading 10 GB and appending 10 GB were 3+ s of configs[1]'s whole process. */
static int64_t count_newlines(const uint8_t *p, int64_t n)   /* eight bytes per step (the reader thread is the wall-clock of a FASTQ run) */
{
	int64_t c = 0, i = 0;
	const uint64_t K = 0x0a0a0a0a0a0a0a0aull, L7 = 0x7f7f7f7f7f7f7f7full;
	for (; i + 32 <= n; i += 32) {
		int k;
		for (k = 0; k < 4; ++k) {
			uint64_t x;
			memcpy(&x, p + i + 8 * k, 8);
			x ^= K;                                               /* bytes equal to '\n' are zero now */
			x = ~(((x & L7) + L7) | x | L7);                      /* 0x80 in every zero byte, exact */
			c += __builtin_popcountll(x);
		}
	}
	for (; i < n; ++i) c += p[i] == '\n';
	return c;
}

typedef struct { const uint8_t *p; int64_t n, c; } nljob_t;
static void *nl_worker(void *a) { nljob_t *j = (nljob_t*)a; j->c = count_newlines(j->p, j->n); return 0; }
static int64_t count_newlines_par(const uint8_t *p, int64_t n)   /* four threads: at 8 GB/s the count of a 16 MiB block was most of what the FASTQ reader thread did */
{
	enum { T = 4 };
	nljob_t job[T]; pthread_t th[T]; int k, started = 0;
	int64_t c = 0;
	if (n < (2 << 20)) return count_newlines(p, n);
	for (k = 0; k < T; ++k) {
		const int64_t o = n / T * k;
		job[k].p = p + o; job[k].n = k == T - 1 ? n - o : n / T; job[k].c = 0;
		if (k > 0) { if (pthread_create(&th[k], 0, nl_worker, &job[k]) == 0) started |= 1 << k; else nl_worker(&job[k]); }
	}
	nl_worker(&job[0]);
	for (k = 1; k < T; ++k) if (started >> k & 1) pthread_join(th[k], 0);
	for (k = 0; k < T; ++k) c += job[k].c;
	return c;
}

static void *pparse_reader(void *arg)
{
	pparse_t *pp = (pparse_t*)arg;
	const int64_t CHUNK = pp->chunk;
	uint8_t *carry = (uint8_t*)xrealloc(0, CHUNK + 16); int64_t n_carry = 0, m_carry = CHUNK + 16, total = 0;
	int last_byte = '\n', eof = 0;
	while (!eof) {
		pjob_t *jb;
		int64_t got = 0, cut;
		pthread_mutex_lock(&pp->mu);
		while (pp->n_queued - pp->consumed >= pp->njob && !pp->stop) pthread_cond_wait(&pp->cv_space, &pp->mu);
		if (pp->stop) { pthread_mutex_unlock(&pp->mu); break; }  /* the consumer goes back to the sequential reader: what is held back here is its to read */
		jb = &pp->job[pp->n_queued % pp->njob];                /* free: taken over by the main thread already */
		pthread_mutex_unlock(&pp->mu);
		jb->stream_off = total - n_carry; jb->failed = 0;
		if (jb->m_in < CHUNK + n_carry + 2) { jb->m_in = CHUNK + n_carry + 2; jb->in = (uint8_t*)xrealloc(jb->in, jb->m_in); }
		memcpy(jb->in, carry, n_carry);
		if (pp->fd >= 0) {                                    /* a plain regular file: four threads copy the block out of the page cache */
			got = rb2_par_pread(pp->fd, jb->in + n_carry, CHUNK, total);
			if (got < CHUNK) eof = 1;
		} else
		while (got < CHUNK) {                                 /* gzread may return short counts on pipes */
			const int64_t r = (int64_t)gzread(pp->fp, jb->in + n_carry + got, (unsigned)(CHUNK - got));
			if (r <= 0) { eof = 1; break; }
			got += r;
		}
		total += got;
		if (got) last_byte = jb->in[n_carry + got - 1];
		jb->n_in = n_carry + got;
		jb->at_eof = eof;
		if (!eof) {                                           /* keep the unfinished last line for the next block */
			if (pp->fastq == 2) {                               /* FASTA: in front of the last line that starts with '>' (0: one record longer than the block) */
				const uint8_t *q = jb->in + jb->n_in;
				cut = 0;
				while (q > jb->in + 1 && (q = (const uint8_t*)memrchr(jb->in + 1, '>', (size_t)(q - jb->in - 1))) != 0) {
					if (q[-1] == '\n') { cut = q - jb->in; break; }
				}
			} else
			for (cut = jb->n_in; cut > 0 && jb->in[cut - 1] != '\n'; --cut);
			if (pp->fastq == 1 && cut > 0) {                         /* ... and, for FASTQ, the lines behind the last multiple of four: a block holds whole records */
				const int64_t nl = count_newlines_par(jb->in, cut);
				int back;
				for (back = (int)(nl & 3); back > 0; --back) for (--cut; cut > 0 && jb->in[cut - 1] != '\n'; --cut);
			}
			n_carry = jb->n_in - cut;                           /* (cut == 0: one line longer than a block -- everything is carried on) */
			if (n_carry > m_carry) { m_carry = n_carry + CHUNK; carry = (uint8_t*)xrealloc(carry, m_carry); }
			memcpy(carry, jb->in + cut, n_carry);
			jb->n_in = cut;
		} else n_carry = 0;
		/* kseq learns about the end of input only from a short refill (kseq.h:70-75, 96-105): an input of k x 16384 bytes
		 * that ends with a newline (or is empty) yields one more, empty, line */
		jb->extra_empty = eof && total % RD_BUF == 0 && last_byte == '\n';
		pthread_mutex_lock(&pp->mu);
		if (jb->n_in > 0 || jb->extra_empty) {
			jb->state = 1; ++pp->n_queued;
			pthread_cond_signal(&pp->cv_work);
		}
		if (eof) pp->eof = 1;
		pthread_cond_broadcast(&pp->cv_done);                  /* (the main thread also waits for blocks to exist) */
		pthread_mutex_unlock(&pp->mu);
	}
	pthread_mutex_lock(&pp->mu);
	pp->carry = carry; pp->n_carry = n_carry; pp->total = total;   /* (freed by the main thread) */
	if (!eof) pp->eof = 1;
	pthread_cond_broadcast(&pp->cv_done);
	pthread_mutex_unlock(&pp->mu);
	return 0;
}

static int usage(int block_len, int max_nodes)
{
	fprintf(stderr, "\nUsage:   ropebwt2-%s [options] <in.fq.gz>\n\n", RB2_VERSION);
	fprintf(stderr, "Options: -l INT     leaf block length [%d]\n", block_len);
	fprintf(stderr, "         -n INT     max number children per internal node [%d]\n", max_nodes);
	fprintf(stderr, "         -s         build BWT in the reverse lexicographical order (RLO)\n");
	fprintf(stderr, "         -r         build BWT in RCLO, overriding -s \n");
	fprintf(stderr, "         -m INT     batch size for multi-string indexing on the GPU; 0 for single-string (CPU) [10g]\n");
	fprintf(stderr, "         -P         accepted for compatibility (the GPU path has no thread switch)\n");
	fprintf(stderr, "         -M INT     accepted for compatibility\n\n");
	fprintf(stderr, "         -i FILE    read existing index in the FMR format from FILE, overriding -s/-r [null]\n");
	fprintf(stderr, "         -L         input in the one-sequence-per-line format\n");
	fprintf(stderr, "         -F         skip forward strand\n");
	fprintf(stderr, "         -R         skip reverse strand\n");
	fprintf(stderr, "         -N         skip sequences containing ambiguous bases\n");
	fprintf(stderr, "         -x INT     cut at ambiguous bases and discard segment with length <INT [0]\n");
	fprintf(stderr, "         -C         cut one base if forward==reverse\n");
	fprintf(stderr, "         -q INT     hard mask bases with QUAL<INT [0]\n\n");
	fprintf(stderr, "         -o FILE    write output to FILE [stdout]\n");
	fprintf(stderr, "         -b         dump the index in the binary FMR format\n");
	fprintf(stderr, "         -d         dump the index in fermi's FMD format\n");
	fprintf(stderr, "         -T         output the index in the Newick format (for debugging)\n\n");
	return 1;
}

/* ---- the batch buffer travels to the device while it is being filled (mr_prefetch): every PF_STEP bytes the reader announces what
 * is final so far.  The device copy reads the host buffer asynchronously, so the buffer must not move: it is reserved at its full
 * size when the batch starts, and an append that would outgrow it (one giant record) cancels the announcements first. ---- */
#define PF_STEP ((size_t)256 << 20)
static struct { size_t sent; int on; int64_t cap; } PF = { 0, 0, 0 };
/* the copies themselves are issued by a thread of their own: a pageable host-to-device copy keeps its caller busy for the time of
 * the copy, and the reader is the wall-clock of a whole run (on the reader's thread the announcements cost configs[1] 0.3 s) */
static struct {
	pthread_t th; pthread_mutex_t mu; pthread_cond_t cv;
	mrope_t *mr; const uint8_t *s; int64_t n, cap; int pending, busy, quit, started;
} UP = { .mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER };

static void *up_worker(void *arg)
{
	(void)arg;
	pthread_mutex_lock(&UP.mu);
	for (;;) {
		while (!UP.pending && !UP.quit) pthread_cond_wait(&UP.cv, &UP.mu);
		if (!UP.pending) break;
		{
			mrope_t *mr = UP.mr; const uint8_t *s = UP.s; const int64_t n = UP.n, cap = UP.cap;
			UP.pending = 0; UP.busy = 1;
			pthread_mutex_unlock(&UP.mu);
			mr_prefetch(mr, s, n, cap);
			pthread_mutex_lock(&UP.mu);
			UP.busy = 0;
			pthread_cond_broadcast(&UP.cv);
		}
	}
	pthread_mutex_unlock(&UP.mu);
	return 0;
}
static void up_drain(void)                                   /* nothing announced is still waiting or being copied */
{
	if (!UP.started) return;
	pthread_mutex_lock(&UP.mu);
	while (UP.pending || UP.busy) pthread_cond_wait(&UP.cv, &UP.mu);
	pthread_mutex_unlock(&UP.mu);
}
static void up_stop(void)
{
	if (!UP.started) return;
	up_drain();
	pthread_mutex_lock(&UP.mu); UP.quit = 1; pthread_cond_broadcast(&UP.cv); pthread_mutex_unlock(&UP.mu);
	pthread_join(UP.th, 0);
	UP.started = 0; UP.quit = 0;
}

static void batch_room(mrope_t *mr, str_t *buf, int64_t m, size_t add)
{
	if (!PF.on) return;
	if (buf->l == 0 && buf->m < (size_t)m + (64 << 20)) { up_drain(); mr_prefetch(mr, 0, 0, 0); str_reserve(buf, (size_t)m + (64 << 20)); PF.sent = 0; }
	if (buf->l + add + 1 > buf->m) { up_drain(); mr_prefetch(mr, 0, 0, 0); PF.sent = (size_t)-1; }      /* it will move: no announcements for the rest of this batch */
}
static void batch_announce(mrope_t *mr, str_t *buf)
{
	if (!PF.on || PF.sent == (size_t)-1 || buf->l < PF.sent + PF_STEP) return;
	if (!UP.started) { UP.started = 1; pthread_create(&UP.th, 0, up_worker, 0); }
	pthread_mutex_lock(&UP.mu);
	UP.mr = mr; UP.s = (const uint8_t*)buf->s; UP.n = (int64_t)buf->l; UP.cap = (int64_t)buf->m; UP.pending = 1;   /* (a later announcement replaces one not yet taken) */
	pthread_cond_broadcast(&UP.cv);
	pthread_mutex_unlock(&UP.mu);
	PF.sent = buf->l;
}

static void flush_batch_now(mrope_t *mr, str_t *buf, int flag, int verbose)
{
	const double c0 = cputime(), r0 = realtime();
	if (getenv("RB2_DUMP_BATCHES")) {                       /* debugging / tests of the readers without a GPU: the batches go to a file, nothing is inserted */
		FILE *fp = fopen(getenv("RB2_DUMP_BATCHES"), "ab");
		const int64_t n = (int64_t)buf->l;
		if (fp) { fwrite(&n, 8, 1, fp); fwrite(buf->s, 1, buf->l, fp); fclose(fp); }
		buf->l = 0;
		return;
	}
	mr_insert_multi(mr, (int64_t)buf->l, (const uint8_t*)buf->s, flag & F_THR);
	mr_wait(mr);                                             /* (the call may return with the GPU still inserting; this line reports the insert, and the CLI overlaps its reading on threads of its own) */
	if (verbose >= 3) fprintf(stderr, "[M::%s] inserted %ld symbols in %.3f sec, %.3f CPU sec\n", "main_ropebwt2", (long)buf->l, realtime() - r0, cputime() - c0);
	buf->l = 0;
}

/* A full batch goes to an inserter thread and the reader gets the other buffer: the GPU inserts batch k while batch k + 1 is
 * read and encoded (the reference inserts where it reads, main.c:238-242; same batches, same order, same calls).
 * RB2_SYNC_INSERT=1 (and RB2_DUMP_BATCHES) keep everything in the caller's thread. */
static struct {
	pthread_t th; pthread_mutex_t mu; pthread_cond_t cv;
	mrope_t *mr; str_t job; int flag, verbose, busy, quit, started;
} AF = { .mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER };

static void *aflush_worker(void *arg)
{
	(void)arg;
	pthread_mutex_lock(&AF.mu);
	for (;;) {
		while (!AF.busy && !AF.quit) pthread_cond_wait(&AF.cv, &AF.mu);
		if (!AF.busy) break;
		pthread_mutex_unlock(&AF.mu);
		flush_batch_now(AF.mr, &AF.job, AF.flag, AF.verbose);
		pthread_mutex_lock(&AF.mu);
		AF.busy = 0;
		pthread_cond_broadcast(&AF.cv);
	}
	pthread_mutex_unlock(&AF.mu);
	return 0;
}

static void flush_wait(void)                                /* every batch handed over so far is in the index */
{
	if (!AF.started) return;
	pthread_mutex_lock(&AF.mu);
	while (AF.busy) pthread_cond_wait(&AF.cv, &AF.mu);
	pthread_mutex_unlock(&AF.mu);
}

static void flush_done(void)
{
	up_stop();
	if (!AF.started) return;
	flush_wait();
	pthread_mutex_lock(&AF.mu);
	AF.quit = 1;
	pthread_cond_broadcast(&AF.cv);
	pthread_mutex_unlock(&AF.mu);
	pthread_join(AF.th, 0);
	free(AF.job.s); AF.job.s = 0; AF.job.l = AF.job.m = 0;
	AF.started = 0; AF.quit = 0;
}

static void flush_batch(mrope_t *mr, str_t *buf, int flag, int verbose)
{
	str_t t;
	if (getenv("RB2_DUMP_BATCHES") || getenv("RB2_SYNC_INSERT")) { flush_batch_now(mr, buf, flag, verbose); return; }
	if (!AF.started) { AF.started = 1; pthread_create(&AF.th, 0, aflush_worker, 0); }
	up_drain();                                                /* what was announced of this batch is on its way before the inserter takes the buffer */
	pthread_mutex_lock(&AF.mu);
	while (AF.busy) pthread_cond_wait(&AF.cv, &AF.mu);       /* the batch before this one */
	t = AF.job; AF.job = *buf; *buf = t;                       /* the reader goes on in the buffer the inserter is done with */
	buf->l = 0; PF.sent = 0;
	AF.mr = mr; AF.flag = flag; AF.verbose = verbose; AF.busy = 1;
	pthread_cond_broadcast(&AF.cv);
	pthread_mutex_unlock(&AF.mu);
}

int main(int argc, char *argv[])
{
	mrope_t *mr = 0;
	FILE *fp_restore = 0;
	reader_t *rd;
	int64_t m = (int64_t)(.97 * 10 * 1024 * 1024 * 1024) + 1;   /* main.c:94 */
	int m_auto = 0;
	int c, i, block_len = ROPE_DEF_BLOCK_LEN, max_nodes = ROPE_DEF_MAX_NODES, verbose = 3, so = MR_SO_IO, min_q = 0, thr_min = -1, min_cut = 0;
	int flag = F_FOR | F_REV | F_THR, ret = 0;
	str_t buf = { 0, 0, 0 };
	double t0 = realtime(), ct, rt;
	FILE *out = stdout;

	while ((c = getopt(argc, argv, "BPNLTFRCtrbdsl:n:m:v:o:i:q:M:x:")) >= 0) {
		switch (c) {
		case 'o': if ((out = fopen(optarg, "wb")) == 0) { fprintf(stderr, "[E::%s] fail to open '%s' for writing\n", __func__, optarg); return 1; } break;
		case 'F': flag &= ~F_FOR; break;
		case 'R': flag &= ~F_REV; break;
		case 'C': flag |= F_ODD; break;
		case 'T': flag |= F_TREE; break;
		case 'b': flag |= F_BIN; break;
		case 't': flag |= F_THR; break;
		case 'L': flag |= F_LINE; break;
		case 'd': flag |= F_RLD; break;
		case 'N': flag |= F_NON; break;
		case 'P': flag &= ~F_THR; break;
		case 'B': fprintf(stderr, "[E::%s] the CRLF output (-B) is not provided by this build\n", __func__); return 1;
		case 's': if (so != MR_SO_RCLO) so = MR_SO_RLO; break;
		case 'r': so = MR_SO_RCLO; break;
		case 'l': block_len = atoi(optarg); break;
		case 'n': max_nodes = atoi(optarg); break;
		case 'v': verbose = atoi(optarg); break;
		case 'q': min_q = atoi(optarg); break;
		case 'M': thr_min = atoi(optarg); break;
		case 'x': min_cut = atoi(optarg); flag |= F_CUTN; break;
		case 'i': {
			FILE *fp = fopen(optarg, "rb");                  /* (main.c:123-127 restores here; the file is read below, once -m is known) */
			if (fp == 0) { fprintf(stderr, "[E::%s] fail to open file '%s'\n", __func__, optarg); return 1; }
			if (fp_restore) fclose(fp_restore);
			fp_restore = fp;
			break; }
		case 'm': {
			char *p; double x;
			if (strcmp(optarg, "auto") == 0) { m_auto = 1; break; }   /* rb2 extension: sized from the device's free memory once the index exists */
			x = strtod(optarg, &p);
			if (*p == 'K' || *p == 'k') x *= 1024;
			else if (*p == 'M' || *p == 'm') x *= 1024 * 1024;
			else if (*p == 'G' || *p == 'g') x *= 1024 * 1024 * 1024;
			m = x ? (int64_t)(x * .97) + 1 : 0;              /* main.c:136 */
			break; }
		default: break;                                      /* unknown options are ignored, as in the reference (main.c:100-137 has no default) */
		}
	}
	if (fp_restore) {                                        /* batch mode builds on the GPU: the run bytes are all it needs (mr_restore_runs) */
		mr = m ? mr_restore_runs(fp_restore) : mr_restore(fp_restore);
		fclose(fp_restore);
		if (mr == 0) return 1;
	}
	if (optind == argc && isatty(fileno(stdin))) return usage(block_len, max_nodes);
	if ((flag & F_CUTN) && m == 0) { fprintf(stderr, "[E::%s] option '-x' cannot be used with '-m0'\n", __func__); return 1; }

	nt6_init();
	if (mr == 0) mr = mr_init(max_nodes, block_len, so);
	if (m_auto) {
		m = (int64_t)(mr_auto_batch_bytes(mr) * .97) + 1;
		if (verbose >= 3) fprintf(stderr, "[M::%s] -m auto: batches of %.1f GiB\n", "main_ropebwt2", m / .97 / 1073741824.0);
	}
	if (thr_min > 0) mr_thr_min(mr, thr_min);
	rd = (reader_t*)calloc(1, sizeof(reader_t));
	rd->fp = optind < argc && strcmp(argv[optind], "-") ? gzopen(argv[optind], "rb") : gzdopen(fileno(stdin), "rb");
	if (rd->fp == 0) { fprintf(stderr, "[E::%s] fail to open the input\n", __func__); return 1; }
	if (m && optind < argc && strcmp(argv[optind], "-") && !getenv("RB2_NO_RESERVE")) {   /* a named file: its size bounds the job (one symbol per byte and strand) */
		struct stat st;
		if (stat(argv[optind], &st) == 0 && S_ISREG(st.st_mode) && st.st_size > (64 << 20)) {
			int64_t c[6], have = 0, a, est = (int64_t)st.st_size * (((flag & F_FOR) ? 1 : 0) + ((flag & F_REV) ? 1 : 0));
			mr_get_c(mr, c);
			for (a = 0; a < 6; ++a) have += c[a];
			mr_reserve(mr, 0, have + est);
		}
	}
	ct = cputime(); rt = realtime();
	if (verbose >= 4) fprintf(stderr, "[M::%s] set up (options, index, device buffers) in %.3f sec\n", "main_ropebwt2", rt - t0);

	{
	enc_cfg_t cfg;
	long pthr = sysconf(_SC_NPROCESSORS_ONLN) - 1;
	cfg.flag = flag; cfg.min_q = min_q; cfg.min_cut = min_cut; cfg.batch = m != 0;
	PF.on = m >= (int64_t)(2 * PF_STEP) && !getenv("RB2_DUMP_BATCHES") && !getenv("RB2_SYNC_INSERT") && !getenv("RB2_NO_PREFETCH");   /* batches worth announcing */
	if (pthr > 16) pthr = 16;
	if (getenv("RB2_PARSE_THREADS")) pthr = atol(getenv("RB2_PARSE_THREADS"));
	if (pthr > 30) pthr = 30;                                 /* (62 block slots, two per thread) */
	{
	int par = 0, need_seq = 1;                              /* 1: -L, blocks of whole lines; 2: FASTQ, blocks of whole four-line records (pjob_encode_fastq) */
	pparse_t pp;
	const uint8_t *fb_mem[64]; int64_t fb_n[64];              /* raw blocks handed back to the sequential reader (FASTQ fallback) */
	memset(&pp, 0, sizeof(pp));
	if (m && pthr > 1) {
		if (flag & F_LINE) par = 1;
		else if (!getenv("RB2_SEQ_FASTX")) {                  /* a file that starts with '@' is taken for four-line FASTQ until a block says otherwise */
			const int c0 = gzgetc(rd->fp);
			if (c0 >= 0) gzungetc(c0, rd->fp);
			if (c0 == '@') par = 2;
			else if (c0 == '>') par = 3;                         /* ... one that starts with '>' for plain FASTA (pjob_encode_fasta) */
		}
	}
	if (par) {
		const int64_t CHUNK = getenv("RB2_PARSE_CHUNK") ? atol(getenv("RB2_PARSE_CHUNK")) : 16 << 20;
		pthread_t *th = (pthread_t*)calloc(pthr, sizeof(pthread_t)), reader;
		int k, fell_back = 0;
		pp.cfg = cfg; pp.njob = (int)pthr * 2 + 2; if (pp.njob > 62) pp.njob = 62;
		pp.job = (pjob_t*)calloc(pp.njob, sizeof(pjob_t));
		pp.fp = rd->fp; pp.chunk = CHUNK; pp.fastq = par == 2 ? 1 : par == 3 ? 2 : 0;
		pp.fd = -1;
		if (optind < argc && strcmp(argv[optind], "-") && gzdirect(rd->fp) == 1 && !getenv("RB2_NO_DIRECT_READ")) {   /* a plain file: bypass zlib */
			struct stat st;
			if (stat(argv[optind], &st) == 0 && S_ISREG(st.st_mode)) pp.fd = open(argv[optind], O_RDONLY);
		}
		pthread_mutex_init(&pp.mu, 0); pthread_cond_init(&pp.cv_work, 0); pthread_cond_init(&pp.cv_done, 0); pthread_cond_init(&pp.cv_space, 0);
		if (par == 1 && pp.fd >= 0 && !getenv("RB2_NO_DIRECT_BLOCKS")) {   /* -L, plain regular file: the workers read their blocks themselves */
			struct stat st;
			if (fstat(pp.fd, &st) == 0 && S_ISREG(st.st_mode)) {
				uint8_t last = '\n';
				pp.direct = 1; pp.fsize = (int64_t)st.st_size;
				if (pp.fsize > 0 && pread(pp.fd, &last, 1, (off_t)(pp.fsize - 1)) != 1) last = 0;
				pp.last_is_nl = last == '\n';
				pp.n_queued = pp.fsize > 0 ? (pp.fsize + CHUNK - 1) / CHUNK : 1;
				pp.eof = 1;
			}
		}
		for (k = 0; k < pthr; ++k) pthread_create(&th[k], 0, pparse_worker, &pp);
		if (!pp.direct) pthread_create(&reader, 0, pparse_reader, &pp);
		for (;;) {                                              /* take over the finished blocks, in order */
			pjob_t *jb;
			size_t done = 0, r0 = 0;
			pthread_mutex_lock(&pp.mu);
			while (pp.consumed >= pp.n_queued && !pp.eof) pthread_cond_wait(&pp.cv_done, &pp.mu);
			if (pp.consumed >= pp.n_queued) { pthread_mutex_unlock(&pp.mu); break; }   /* end of input, everything taken over */
			jb = &pp.job[pp.consumed % pp.njob];
			while (jb->state != 2) pthread_cond_wait(&pp.cv_done, &pp.mu);
			if (pp.fastq && jb->failed) {                       /* not strict four-line FASTQ from here on: stop the reader, keep the raw blocks */
				pp.stop = 1; fell_back = 1;
				pthread_cond_broadcast(&pp.cv_space);
				pthread_mutex_unlock(&pp.mu);
				break;
			}
			pthread_mutex_unlock(&pp.mu);
			while (done < jb->out.l) {                          /* same flush points as the sequential loop: after the record that fills the batch */
				size_t lo = r0, hi = jb->n_rec;                 /* first record whose end reaches the threshold */
				const int64_t need = m - (int64_t)buf.l;
				while (lo < hi) { const size_t mid = (lo + hi) >> 1; if ((int64_t)(jb->rec_end[mid] - done) >= need) hi = mid; else lo = mid + 1; }
				if (lo == jb->n_rec) { batch_room(mr, &buf, m, jb->out.l - done); str_append(&buf, jb->out.s + done, jb->out.l - done); done = jb->out.l; batch_announce(mr, &buf); }
				else {
					batch_room(mr, &buf, m, jb->rec_end[lo] - done);
					str_append(&buf, jb->out.s + done, jb->rec_end[lo] - done); done = jb->rec_end[lo]; r0 = lo + 1;
					flush_batch(mr, &buf, flag, verbose);
				}
			}
			pthread_mutex_lock(&pp.mu);
			jb->state = 0; ++pp.consumed;
			pthread_cond_broadcast(&pp.cv_space);
			pthread_mutex_unlock(&pp.mu);
		}
		if (!pp.direct) pthread_join(reader, 0);
		pthread_mutex_lock(&pp.mu); pp.closing = 1; pthread_cond_broadcast(&pp.cv_work); pthread_cond_broadcast(&pp.cv_space); pthread_mutex_unlock(&pp.mu);
		for (k = 0; k < pthr; ++k) pthread_join(th[k], 0);
		free(th);
		if (pp.fd >= 0) close(pp.fd);
		need_seq = fell_back;
		if (getenv("RB2_PARSE_TRACE")) fprintf(stderr, "[M::%s] %ld blocks of %s parsed by %ld threads%s\n", "main_ropebwt2", (long)pp.consumed, par == 2 ? "four-line FASTQ records" : par == 3 ? "FASTA records" : pp.direct ? "lines (read by the workers)" : "lines", pthr, fell_back ? ", then the sequential reader" : "");
		if (fell_back) {                                        /* the blocks from the failed one on, what the reader held back, then the stream itself */
			int64_t q;
			int n = 0;
			for (q = pp.consumed; q < pp.n_queued; ++q) { pjob_t *jb = &pp.job[q % pp.njob]; fb_mem[n] = jb->in; fb_n[n] = jb->n_in; ++n; }
			if (pp.n_carry > 0) { fb_mem[n] = pp.carry; fb_n[n] = pp.n_carry; ++n; }
			rd->mem = fb_mem; rd->mem_n = fb_n; rd->nmem = n; rd->imem = 0; rd->mem_off = 0;
			rd->pos = pp.job[pp.consumed % pp.njob].stream_off;
			rd->beg = rd->end = 0; rd->eof = 0; rd->last = 0;
			if (pp.fd >= 0) gzseek(rd->fp, (z_off_t)pp.total, SEEK_SET);   /* the reader thread read the file itself: the stream goes on where it stopped */
			if (verbose >= 3) fprintf(stderr, "[M::%s] the input is not %s from byte %ld on: sequential reader\n", "main_ropebwt2", par == 2 ? "four-line FASTQ" : "plain FASTA", (long)rd->pos);
		}
	}
	if (need_seq)
	while ((flag & F_LINE ? read_line_record(rd) : read_fastx_record(rd)) >= 0) {
		uint8_t *s = (uint8_t*)rd->seq.s;
		int l = prepare_record(&cfg, s, (int)rd->seq.l, rd->qual.s, (int)rd->qual.l);
		if (l < 0) continue;
		if (m) {
			batch_room(mr, &buf, m, 2 * ((size_t)l + 1));
			append_strands(&cfg, s, l, &buf);
			if ((int64_t)buf.l >= m) flush_batch(mr, &buf, flag, verbose);
			else batch_announce(mr, &buf);
		} else {
			if (flag & F_FOR) mr_insert1(mr, s);
			if (flag & F_REV) { revcomp_in_place(s, l); mr_insert1(mr, s); }
		}
	}
	if (par) {
		int k;
		for (k = 0; k < pp.njob; ++k) { free(pp.job[k].in); free(pp.job[k].out.s); free(pp.job[k].rec_end); }
		free(pp.job); free(pp.carry);
		pthread_mutex_destroy(&pp.mu); pthread_cond_destroy(&pp.cv_work); pthread_cond_destroy(&pp.cv_done); pthread_cond_destroy(&pp.cv_space);
	}
	}
	}
	if (m && buf.l) flush_batch(mr, &buf, flag, verbose);
	flush_done();
	if (verbose >= 3) {
		int64_t cc[6];
		fprintf(stderr, "[M::%s] constructed FM-index in %.3f sec, %.3f CPU sec\n", "main_ropebwt2", realtime() - rt, cputime() - ct);
		mr_get_c(mr, cc);
		fprintf(stderr, "[M::%s] symbol counts: ($, A, C, G, T, N) = (%ld, %ld, %ld, %ld, %ld, %ld)\n", "main_ropebwt2",
				(long)cc[0], (long)cc[1], (long)cc[2], (long)cc[3], (long)cc[4], (long)cc[5]);
	}
	{ const double tf = realtime(); free(buf.s); gzclose(rd->fp); free(rd->seq.s); free(rd->qual.s); free(rd);
	  if (verbose >= 4) fprintf(stderr, "[M::%s] batch buffers and reader released in %.3f sec\n", "main_ropebwt2", realtime() - tf); }

	if (out != stdout) { fflush(stdout); if (dup2(fileno(out), fileno(stdout)) < 0) return 1; }   /* mr_print_tree writes to stdout */
	{ static char obuf[4 << 20]; fflush(stdout); setvbuf(stdout, obuf, _IOFBF, sizeof(obuf)); }   /* .fmr dumps are millions of small fwrites */
	if (flag & F_BIN) {
		const double td0 = realtime();
		mr_dump(mr, stdout);                                    /* a regular file: leaf records straight from the device's run bytes, no host trees (mrope.c) */
		fflush(stdout);
		if (verbose >= 3) fprintf(stderr, "[M::%s] BWT off the device and written as .fmr in %.3f sec\n", "main_ropebwt2", realtime() - td0);
	}
	else if (flag & F_TREE) mr_print_tree(mr);
	else {
		/* .fmd: the Elias-delta coding of the run stream is spread over worker threads (fmd.c: speculative segments + one
		 * stitching pass; RB2_FMD_THREADS=0 for the plain sequential writer path with one worker) */
		rb2_fmdp_t *fmdp = 0;
		if (flag & F_RLD) {
			long nt = sysconf(_SC_NPROCESSORS_ONLN) - 1;
			if (nt > 24) nt = 24;
			if (getenv("RB2_FMD_THREADS")) nt = atol(getenv("RB2_FMD_THREADS"));
			fmdp = rb2_fmdp_init(nt < 1 ? 1 : nt > 64 ? 64 : (int)nt, getenv("RB2_FMD_SEGMENT") ? atol(getenv("RB2_FMD_SEGMENT")) : 0);
			{ int64_t cc[6], tot = 0; int a; mr_get_c(mr, cc); for (a = 0; a < 6; ++a) tot += cc[a]; rb2_fmdp_expect(fmdp, tot); }
			if (!getenv("RB2_FMD_NO_STREAM")) {                 /* -o file / a redirected stdout: the index goes out while it is encoded */
				fflush(stdout);
				const off_t at = ftello(stdout);
				if (at >= 0 && rb2_fmdp_set_output(fmdp, fileno(stdout), (int64_t)at) == 0 && verbose >= 4)
					fprintf(stderr, "[M::%s] the .fmd is written while it is encoded\n", "main_ropebwt2");
			}
		}
		const double ts0 = realtime();
		mr_stream_runs(mr, emit_runs, fmdp);                   /* the reference walks mr_itr_next_block here (main.c:288-305) */
		if (fmdp) {
			int64_t cc[7];
			const double ts1 = realtime();
			rb2_fmd_t *fmd = rb2_fmdp_finish(fmdp);
			if (verbose >= 3) fprintf(stderr, "[M::%s] BWT streamed off the device and run-length coded in %.3f sec (+ %.3f sec to finish and index)\n", "main_ropebwt2", ts1 - ts0, realtime() - ts1);
			rb2_fmd_counts(fmd, cc);
			fprintf(stderr, "[M::%s] rld: (tot, $, A, C, G, T, N) = (%ld, %ld, %ld, %ld, %ld, %ld, %ld)\n", "main_ropebwt2",
					(long)cc[0], (long)cc[1], (long)cc[2], (long)cc[3], (long)cc[4], (long)cc[5], (long)cc[6]);
			{ const double tw = realtime(); double tw1;
			  if (rb2_fmd_write(fmd, stdout) != 0) { fprintf(stderr, "[E::%s] failed to write the index\n", "main_ropebwt2"); ret = 1; }
			  tw1 = realtime();
			  rb2_fmd_destroy(fmd);
			  if (verbose >= 4) fprintf(stderr, "[M::%s] rest of the .fmd written in %.3f sec, encoder released in %.3f sec\n", "main_ropebwt2", tw1 - tw, realtime() - tw1); }
		} else putchar('\n');
	}
	fflush(stdout);
	{ const double td = realtime(); mr_destroy(mr); if (verbose >= 4) fprintf(stderr, "[M::%s] index and device released in %.3f sec\n", "main_ropebwt2", realtime() - td); }
	fprintf(stderr, "[M::%s] Version: %s\n[M::%s] CMD:", "main", RB2_VERSION, "main");
	for (i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
	fprintf(stderr, "\n[M::%s] Real time: %.3f sec; CPU: %.3f sec\n", "main", realtime() - t0, cputime());
	return ret;
}
