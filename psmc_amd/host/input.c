/* input.c -- .psmcfa reader and bootstrap resampling.
 *
 * A .psmcfa is FASTA/FASTQ-like; each sequence character is one 100-bp bin:
 * T/A/C/G/0 = no heterozygote, K/M/R/S/W/Y/1 = at least one, anything else =
 * missing (lh3/psmc cli.c:15-32; utils/fq2psmcfa.c:114-127 writes T/K/N).
 * Record parsing follows kseq.h:172-223 character by character: a record
 * starts at '>' or '@', its name ends at the first white space, the sequence
 * is every isgraph() character up to the next '>', '@' or '+', and a '+' opens
 * a quality block of the same length that is skipped. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include "psmc_host.h"

uint8_t psmc_symbol_of(unsigned char c)
{
	switch (c) {
	case 'T': case 'A': case 'C': case 'G': case 't': case 'a': case 'c': case 'g': case '0': return 0;
	case 'K': case 'M': case 'R': case 'S': case 'W': case 'Y':
	case 'k': case 'm': case 'r': case 's': case 'w': case 'y': case '1': return 1;
	default: return 2;
	}
}

typedef struct { gzFile fp; unsigned char buf[1 << 16]; int pos, len, eof; } reader;
static int rd_getc(reader *r)
{
	if (r->pos >= r->len) {
		if (r->eof) return -1;
		r->len = gzread(r->fp, r->buf, sizeof r->buf);
		r->pos = 0;
		if (r->len < (int)sizeof r->buf) r->eof = 1;
		if (r->len <= 0) { r->len = 0; return -1; }
	}
	return r->buf[r->pos++];
}

static void push_segment(psmc_input *in, char *name, uint8_t *sym, int32_t L, int32_t called, int32_t het)
{
	in->seg = (psmc_segment *)realloc(in->seg, sizeof(psmc_segment) * (size_t)(in->n_seg + 1));
	psmc_segment *s = &in->seg[in->n_seg++];
	s->name = name; s->sym = sym; s->L = L; s->L_called = called; s->n_het = het;
}

static void recount(psmc_input *in)
{
	in->sum_called = in->sum_het = 0; /* cli.c:133-137 */
	for (int i = 0; i < in->n_seg; ++i) { in->sum_called += in->seg[i].L_called; in->sum_het += in->seg[i].n_het; }
}

int psmc_input_read(const char *path, psmc_input *in)
{
	reader *r = (reader *)calloc(1, sizeof(reader));
	memset(in, 0, sizeof(*in));
	r->fp = strcmp(path, "-") ? gzopen(path, "r") : gzdopen(fileno(stdin), "r");
	if (!r->fp) { free(r); return -1; }
	int c, pending = 0; /* pending = header character already consumed */
	for (;;) {
		if (!pending) {
			while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {}
			if (c == -1) break;
		}
		pending = 0;
		/* name: up to the first white space; the rest of the header line is a comment */
		size_t nl = 0, nm = 64;
		char *name = (char *)malloc(nm);
		while ((c = rd_getc(r)) != -1 && !isspace(c)) {
			if (nl + 2 > nm) name = (char *)realloc(name, nm *= 2);
			name[nl++] = (char)c;
		}
		name[nl] = 0;
		if (c == -1 && nl == 0) { free(name); break; }
		if (c != '\n') while ((c = rd_getc(r)) != -1 && c != '\n') {}
		/* sequence */
		size_t L = 0, cap = 1 << 16;
		uint8_t *sym = (uint8_t *)malloc(cap);
		int32_t called = 0, het = 0;
		while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
			if (!isgraph(c)) continue;
			if (L + 1 > cap) sym = (uint8_t *)realloc(sym, cap *= 2);
			const uint8_t v = psmc_symbol_of((unsigned char)c);
			sym[L++] = v;
			if (v < 2) { ++called; if (v == 1) ++het; }
		}
		if (c == '>' || c == '@') pending = 1;
		if (c == '+') { /* FASTQ: skip the '+' line and as many quality characters as bases */
			while ((c = rd_getc(r)) != -1 && c != '\n') {}
			size_t q = 0;
			if (c != -1) while ((c = rd_getc(r)) != -1 && q < L) if (c >= 33 && c <= 127) ++q;
			if (q != L) { free(name); free(sym); break; } /* truncated quality: kseq_read returns -2, the loop ends */
		}
		if (L > 0x7fffffff) { free(name); free(sym); gzclose(r->fp); free(r); return -2; }
		push_segment(in, name, sym, (int32_t)L, called, het);
	}
	gzclose(r->fp);
	free(r);
	recount(in);
	return 0;
}

/* -b: draw segments with replacement until the total length is as close as
 * possible to the original (aux.c:8-47); same drand48 stream, same accept rule. */
int psmc_input_resample_idx(const psmc_input *in, int32_t **idx_out)
{	/* the draw itself: which segments, in which order (the multiset psmc_hip_select / psmc_hip_estep_batch take) */
	int64_t have = 0, want = 0;
	int n = 0, cap = in->n_seg + 16;
	int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
	for (int i = 0; i < in->n_seg; ++i) want += in->seg[i].L;
	for (;;) {
		const int pick = (int)(in->n_seg * drand48());
		const psmc_segment *s = &in->seg[pick];
		const int missing = (int)(want - have);          /* tmp1: still to fill   (int like aux.c:16) */
		const int excess = (int)(have + s->L - want);    /* tmp2: overshoot if taken */
		if (excess <= 0 || (excess > 0 && missing > 0 && excess < missing)) {
			if (n == cap) idx = (int32_t *)realloc(idx, sizeof(int32_t) * (size_t)(cap *= 2));
			idx[n++] = pick;
			have += s->L;
		}
		if (missing >= 0 && excess >= 0) break;
	}
	*idx_out = idx;
	return n;
}

void psmc_input_resample(psmc_input *in)
{
	int32_t *idx = 0;
	const int n = psmc_input_resample_idx(in, &idx);
	psmc_input out;
	memset(&out, 0, sizeof out);
	for (int i = 0; i < n; ++i) {
		const psmc_segment *s = &in->seg[idx[i]];
		uint8_t *sym = (uint8_t *)malloc(s->L ? (size_t)s->L : 1);
		memcpy(sym, s->sym, (size_t)s->L);
		push_segment(&out, strdup(s->name), sym, s->L, s->L_called, s->n_het);
	}
	free(idx);
	psmc_input_free(in);
	*in = out;
	recount(in);
}

void psmc_input_free(psmc_input *in)
{
	for (int i = 0; i < in->n_seg; ++i) { free(in->seg[i].name); free(in->seg[i].sym); }
	free(in->seg);
	memset(in, 0, sizeof(*in));
}
