/* main.c -- the `psmc` executable: lh3/psmc's command line (main.c:6-34) with
 * the E-step bound to libpsmc_hip.so.  There is no CPU E-step in this binary:
 * without a visible AMD GPU it exits with an error.
 *   PSMC_HIP_MODE=exact (default: .psmc byte-identical to the reference) | fast
 *   PSMC_HIP_DEVICE=<index>          one GPU
 *   PSMC_HIP_DEVICES=<i>,<j>,...     the segments of every E-step sharded over these GPUs (psmc_hip_group_*: LPT
 *                                    partition, one RCCL all-reduce of the statistics per EM iteration in fast mode,
 *                                    ordered per-segment sum in exact mode -- the output does not depend on the list) */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_hip.h"
#include "hipbe.h"

typedef struct { const char *path; psmc_input *in; int rc; } prefetch_job;
static void *prefetch_input(void *arg) { prefetch_job *j = (prefetch_job *)arg; j->rc = psmc_input_read(j->path, j->in); return 0; }

static int mode_is_fast(void) { const char *s = getenv("PSMC_HIP_MODE"); return s && strcmp(s, "fast") == 0; }

int main(int argc, char *argv[])
{
	psmc_options o;
	psmc_options_default(&o);
	if (psmc_options_parse(&o, argc, argv)) { psmc_options_free(&o); return 1; }
	/* number of states is known only after the pattern (or -i file) is read: peek at it */
	psmc_pattern pat;
	int n_states = 0;
	if (o.param_file) {
		FILE *fp = fopen(o.param_file, "r"); char str[256];
		if (fp && fscanf(fp, "%255s", str) == 1 && psmc_pattern_parse(str, &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
		if (fp) fclose(fp);
	} else if (psmc_pattern_parse(o.pattern_text ? o.pattern_text : "4+5*3+4", &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
	if (n_states < 1) { fprintf(stderr, "psmc: malformed pattern\n"); return 1; }
	if (n_states > PSMC_HIP_MAX_STATES) { /* the reference has no limit at all (khmm.c:10-23 allocates for any n); the wide exact kernels keep one thread per state in a work-group */
		fprintf(stderr, "psmc: the pattern gives %d hidden states; this MI355X build supports at most %d. "
		        "Use a coarser pattern, or the reference binary for this run.\n", n_states, PSMC_HIP_MAX_STATES);
		psmc_options_free(&o);
		return 2;
	}
	if (n_states > 128 && mode_is_fast())
		fprintf(stderr, "psmc: %d hidden states: the fast kernels stop at 128, every E-step of this run uses the exact ones\n", n_states);
	const char *mode_s = getenv("PSMC_HIP_MODE"), *dev_s = getenv("PSMC_HIP_DEVICE");
	int mode = (mode_s && strcmp(mode_s, "fast") == 0) ? PSMC_HIP_MODE_FAST : PSMC_HIP_MODE_EXACT;
	if ((o.decode || o.print_prob || o.cnt_file) && mode == PSMC_HIP_MODE_FAST) {
		fprintf(stderr, "psmc: decoding needs the exact forward/backward tables; using PSMC_HIP_MODE=exact\n");
		mode = PSMC_HIP_MODE_EXACT;
	}
	{ /* the O(N) objective goes with the fast E-step unless asked otherwise */
		const char *fm = getenv("PSMC_FAST_MSTEP");
		o.fast_mstep = fm ? atoi(fm) != 0 : (mode == PSMC_HIP_MODE_FAST);
	}
	/* the factored E-step goes with the O(N) objective: fast mode (PSMC_FACTORED=0 keeps the full counts) */
	const char *fs = getenv("PSMC_FACTORED"), *devs = getenv("PSMC_HIP_DEVICES");
	const int use_factored = o.fast_mstep && mode == PSMC_HIP_MODE_FAST && n_states <= 128 && !(fs && atoi(fs) == 0);
	/* The input is read on a thread of its own while the device comes up: 0.35 s for a 30 M-bin genome beside 0.4 s of HIP start-up, of a
	 * program that takes 1.0 s in all in fast mode (profiles/r06_fast_after_exact.txt).  psmc_run_begin takes the result over. */
	pthread_t rd_tid;
	prefetch_job pj = {o.in_file, (psmc_input *)calloc(1, sizeof(psmc_input)), 0};
	/* (not from stdin: a device that does not come up must say so at once, not after the pipe has closed) */
	const int rd_started = pj.in && o.in_file && strcmp(o.in_file, "-") != 0 && pthread_create(&rd_tid, 0, prefetch_input, &pj) == 0;
	psmc_estep_backend be;
	const int rc = psmc_hipbe_create(&be, n_states, mode, use_factored, devs, dev_s ? atoi(dev_s) : 0);
	if (rd_started) { pthread_join(rd_tid, 0); o.prefetched = pj.in; o.prefetch_rc = pj.rc; }
	else free(pj.in);
	if (rc) {
		fprintf(stderr, "psmc: cannot start the MI355X E-step (%s); this build has no CPU path\n", psmc_hip_strerror(rc));
		psmc_options_free(&o);
		return 2;
	}
	int status = psmc_run(&o, &be);
	be.destroy(be.self);
	psmc_options_free(&o);
	return status;
}
