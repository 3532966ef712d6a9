/* boot_main.c -- the `psmc_boot` executable: all bootstrap replicates of lh3/psmc's README:57-62 in one process on
 * the MI355X(s) of the node.
 *
 *   psmc_boot -R <replicates> [-S <first seed>] -O <out pattern with %d> [--main <out.psmc> --main-input <in.psmcfa>]
 *             -- <psmc options> split.psmcfa
 *
 * e.g.  psmc_boot -R 100 -S 1 -O round-%d.psmc -- -N25 -t15 -r5 -p "4+25*2+4+6" split.psmcfa
 * writes what  for r in 0..99: PSMC_SEED=$((1+r)) psmc -N25 -t15 -r5 -b -p ... -o round-$r.psmc split.psmcfa  would
 * (-b is implied).  PSMC_HIP_MODE=exact|fast and PSMC_FAST_MSTEP as for psmc; PSMC_HIP_DEVICES=0,1,.. (a device may be
 * listed several times = as many contexts on it; default: all visible devices, three times each in fast mode); OMP_NUM_THREADS bounds
 * the M-step threads (default: the processors the control group's CPU quota allows, less the device threads); PSMC_TIMING=1 prints
 * per-iteration times to stderr.  A replicate's M-step starts when the batch reports its statistics final (psmc_hip_estep_batch_cb)
 * and runs under the rest of the batch.
 *
 * --main / --main-input: the whole workflow of the reference's README:49-62 as ONE job -- the un-resampled main run
 * (`psmc <psmc options> -o out.psmc in.psmcfa`, on the unsplit input) runs on a thread of its own BESIDE the replicates, on the
 * first device; its 3.3 s E-steps -- the critical path of one wave over the longest chromosome -- hide behind the batch's 7 s ones
 * instead of preceding them.  Exact mode: the main run's context is masked to PSMC_BOOT_MAIN_CUS compute units (default 32) and the
 * batch to the others (psmc_hip_set_cu_range), so that the two never share a SIMD and the batch sizes its launches for its share.
 * The mask's bits go round the XCDs, then round the four shader engines of an XCD, and the dispatcher deals a launch evenly over
 * the ENGINES: only multiples of 32 leave every engine the same number of units.  Measured with the main run alive (100 replicates
 * of a 30 M-bin genome, profiles/r05_boot_schedule.txt): batch alone 6.75 s per EM iteration; with the main run on 32 masked
 * units 7.27 s, the main run 3.3 s per E-step (alone: 3.09); on 24 units (engines left 7 / 7 / 7 / 8 units) 9.4 s; without masks
 * (round 5's A/B: the batch kept entry slots free instead; removed in round 6) 10 s -- the main run's resident waves
 * keep the batch's work-groups, which need a whole SIMD's registers, out of their compute units, and a launch that has to place one
 * work-group late lasts twice as long.  The main output is byte-identical to `psmc`'s, the replicates to a run without --main
 * (tests/test_host_cli.py).
 * There is no CPU E-step in this binary. */
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"
#include "psmc_hip.h"
#include "hipbe.h"

#define MAX_DEV 64
typedef struct {
	int n_dev, n_states, n_rep; psmc_hip_ctx *ctx[MAX_DEV]; int dev_id[MAX_DEV], main_dev;
	int64_t want_bins[MAX_DEV];      /* what bb_reserve was asked for: the second call, when the main run is over, asks again */
	psmc_estep_backend *be_main;     /* the main run's backend, so that its context (and its 31 GB of tables) can go the moment the run is over */
	int main_gone;
} hip_bb;

static int bb_load(void *self, int dev, int n_seg, const uint8_t *const *sym, const int32_t *L)
{
	return psmc_hip_load_segments(((hip_bb *)self)->ctx[dev], n_seg, sym, L);
}
/* exact mode: take the batch's table memory before the first EM iteration, sized for the replicates as drawn (the driver clears what
 * it hands out: seconds for 250 GB) -- part of the set-up, not of an iteration */
static int bb_reserve(void *self, int dev, int64_t table_bins)
{
	((hip_bb *)self)->want_bins[dev] = table_bins;
	return psmc_hip_reserve_batch_tables(((hip_bb *)self)->ctx[dev], table_bins);
}
/* the main run that shared device main_dev is over: its compute units go back to the batch */
static void bb_main_done(void *self, int d)
{
	hip_bb *h = (hip_bb *)self;
	if (h->dev_id[d] != h->main_dev) return;
	(void)psmc_hip_set_cu_range(h->ctx[d], 0, 0);
	/* ... and its memory: the main run's context goes now (the first device thread to get here does it), and this batch context asks for its
	 * tables again -- the library adds what has become free as a second chunk (psmc_hip_reserve_batch_tables), so that the iterations still to
	 * come need a launch fewer (round 5 kept five launches to the end: 12 s of 197) */
	if (h->be_main && !__atomic_exchange_n(&h->main_gone, 1, __ATOMIC_ACQ_REL) && h->be_main->destroy) { h->be_main->destroy(h->be_main->self); h->be_main->destroy = 0; }
	if (h->want_bins[d] > 0) (void)psmc_hip_reserve_batch_tables(h->ctx[d], h->want_bins[d]);
}
static int bb_estep_batch(void *self, int dev, int n_rep, const double *a, const double *e, const double *a0,
                          const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL,
                          void (*done)(void *user, int n_done, const int32_t *pos), void *user)
{
	return psmc_hip_estep_batch_cb(((hip_bb *)self)->ctx[dev], n_rep, a, e, a0, sel_off, sel_idx, A, sums, E, LL, done, user);
}
static const char *bb_error(void *self, int dev) { return psmc_hip_last_error(((hip_bb *)self)->ctx[dev]); }
static void bb_destroy(void *self) { hip_bb *h = (hip_bb *)self; for (int d = 0; d < h->n_dev; ++d) psmc_hip_destroy(h->ctx[d]); }

static void usage(void)
{
	fprintf(stderr, "Usage: psmc_boot -R <replicates> [-S <first seed>] -O <output pattern with %%d> [--main <out.psmc> --main-input <unsplit.psmcfa>]\n"
	                "                 -- <psmc options> input.psmcfa\n"
	                "       (replicate r = `PSMC_SEED=<seed+r> psmc -b <psmc options> -o <pattern %% r>`; --main: `psmc <psmc options> -o <out.psmc>\n"
	                "        <unsplit.psmcfa>` beside them in the same job; PSMC_HIP_MODE, PSMC_HIP_DEVICES, PSMC_BOOT_MAIN_CUS)\n");
}

int main(int argc, char *argv[])
{
	int n_rep = 0, i = 1;
	long seed0 = 1;
	const char *pattern = 0, *main_out = 0, *main_in = 0;
	{ /* before the first HIP call: the main run's stream and the batch's must not share a hardware queue (a larger value the user exported stands) */
		const char *q = getenv("GPU_MAX_HW_QUEUES");
		if (!q || atoi(q) < 16) setenv("GPU_MAX_HW_QUEUES", "16", 1);
	}
	for (; i < argc; ++i) {
		if (!strcmp(argv[i], "--")) { ++i; break; }
		if (!strcmp(argv[i], "-R") && i + 1 < argc) n_rep = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-S") && i + 1 < argc) seed0 = atol(argv[++i]);
		else if (!strcmp(argv[i], "-O") && i + 1 < argc) pattern = argv[++i];
		else if (!strcmp(argv[i], "--main") && i + 1 < argc) main_out = argv[++i];
		else if (!strcmp(argv[i], "--main-input") && i + 1 < argc) main_in = argv[++i];
		else { usage(); return 1; }
	}
	if (n_rep < 1 || !pattern || i >= argc || (main_out != 0) != (main_in != 0)) { usage(); return 1; }
	psmc_options o, om;
	psmc_options_default(&o); psmc_options_default(&om);
	{ /* psmc's own option parser on the rest of the command line (twice with --main: the main run's copy) */
		char **av = (char **)malloc(sizeof(char *) * (size_t)(argc - i + 2));
		av[0] = argv[0];
		for (int k = i; k < argc; ++k) av[k - i + 1] = argv[k];
		int rc = psmc_options_parse(&o, argc - i + 1, av);
		if (rc == 0 && main_out) rc = psmc_options_parse(&om, argc - i + 1, av);
		free(av);
		if (rc) { psmc_options_free(&o); psmc_options_free(&om); return 1; }
	}
	o.bootstrap = 1;
	if (main_out) { /* `psmc <psmc options> -o main_out main_in`: no -b */
		om.bootstrap = 0;
		free(om.in_file); om.in_file = strdup(main_in);
		free(om.out_file); om.out_file = strdup(main_out);
		if (om.decode || om.cnt_file || om.print_prob || om.simulate) {
			fprintf(stderr, "psmc_boot: --main shares its psmc options with the replicates: decoding / simulation options (-d -D -s -c -S) cannot be given here\n");
			psmc_options_free(&o); psmc_options_free(&om);
			return 1;
		}
	}
	psmc_pattern pat;
	int n_states = 0;
	if (o.param_file) {
		FILE *fp = fopen(o.param_file, "r"); char str[256];
		if (fp && fscanf(fp, "%255s", str) == 1 && psmc_pattern_parse(str, &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
		if (fp) fclose(fp);
	} else if (psmc_pattern_parse(o.pattern_text ? o.pattern_text : "4+5*3+4", &pat) == 0) { n_states = pat.n_states; psmc_pattern_free(&pat); }
	if (n_states < 1) { fprintf(stderr, "psmc_boot: malformed pattern\n"); return 1; }
	const char *mode_s = getenv("PSMC_HIP_MODE"), *devs = getenv("PSMC_HIP_DEVICES"), *fm = getenv("PSMC_FAST_MSTEP");
	const int mode = (mode_s && strcmp(mode_s, "fast") == 0) ? PSMC_HIP_MODE_FAST : PSMC_HIP_MODE_EXACT;
	o.fast_mstep = fm ? atoi(fm) != 0 : (mode == PSMC_HIP_MODE_FAST);
	om.fast_mstep = o.fast_mstep;
	hip_bb h;
	memset(&h, 0, sizeof h);
	h.n_states = n_states;
	int list[MAX_DEV], n_list = 0;
	if (devs && *devs) {
		char *dup = strdup(devs);
		for (char *t = strtok(dup, ","); t && n_list < MAX_DEV; t = strtok(0, ",")) list[n_list++] = atoi(t);
		free(dup);
	} else {
		/* all visible devices.  Fast mode: THREE contexts per device, each with its own tables and driver thread -- a fast
		 * E-step is a chain of short launches with host decisions in between, and the other threads' replicates fill the
		 * gaps of the first (100 replicates of a 30 M-bin genome, per EM iteration: one context 1.19 s, two 0.69, three 0.585,
		 * four 0.72, six 0.66: profiles/r05_fast_contexts.txt).  Exact mode packs the device with one batch already, and
		 * wants all of the memory for its tables. */
		const int nd = psmc_hip_device_count(), per = mode == PSMC_HIP_MODE_FAST ? 3 : 1;
		for (int d = 0; d < nd && n_list + per <= MAX_DEV; ++d)
			for (int k = 0; k < per; ++k) list[n_list++] = d;
	}
	if (n_list > n_rep) n_list = n_rep;
	if (n_list < 1) { fprintf(stderr, "psmc_boot: no MI355X visible; this build has no CPU path\n"); psmc_options_free(&o); return 2; }
	for (int d = 0; d < n_list; ++d) {
		const int rc = psmc_hip_create(&h.ctx[d], n_states, list[d], mode);
		if (rc) {
			fprintf(stderr, "psmc_boot: cannot start the E-step on device %d (%s); this build has no CPU path\n", list[d], psmc_hip_strerror(rc));
			for (int k = 0; k < d; ++k) psmc_hip_destroy(h.ctx[k]);
			psmc_options_free(&o);
			return 2;
		}
		h.dev_id[d] = list[d];
		h.n_dev = d + 1;
	}
	h.main_dev = -1;
	h.n_rep = n_rep;
	const char *fs = getenv("PSMC_FACTORED");
	psmc_batch_backend bb = {&h, h.n_dev, bb_load, bb_estep_batch, bb_error, bb_destroy, mode == PSMC_HIP_MODE_FAST && !(fs && atoi(fs) == 0), bb_reserve, 0};
	/* the main run: a context of its own on the first device, begun (header, input, RD 0, tables) before the batch sizes its tables */
	psmc_estep_backend be_main;
	psmc_run_state *main_run = 0;
	memset(&be_main, 0, sizeof be_main);
	if (main_out) {
		const int use_factored = om.fast_mstep && mode == PSMC_HIP_MODE_FAST && n_states <= 128 && !(fs && atoi(fs) == 0);
		int rc = psmc_hipbe_create(&be_main, n_states, mode, use_factored, 0, list[0]);
		if (rc == 0 && mode == PSMC_HIP_MODE_EXACT) { /* split the first device: [0, m) main run, [m, all) the batch contexts on it */
			h.main_dev = list[0]; bb.main_done = bb_main_done;
			const char *ms = getenv("PSMC_BOOT_MAIN_CUS");
			const int cus = psmc_hip_device_cus(list[0]), m = ms ? atoi(ms) : 32;
			if (m > 0 && m < cus) {
				rc = psmc_hip_set_cu_range(psmc_hipbe_ctx(&be_main), 0, m);
				for (int d = 0; d < h.n_dev && rc == 0; ++d)
					if (list[d] == list[0]) rc = psmc_hip_set_cu_range(h.ctx[d], m, cus - m);
			}
		}
		if (rc) {
			fprintf(stderr, "psmc_boot: cannot set up the main run on device %d (%s)\n", list[0], psmc_hip_strerror(rc));
			if (be_main.destroy) be_main.destroy(be_main.self);
			bb.destroy(bb.self); psmc_options_free(&o); psmc_options_free(&om);
			return 2;
		}
		h.be_main = &be_main;
		main_run = psmc_run_begin(&om, &be_main);
		if (main_run && (rc = psmc_hip_reserve_tables(psmc_hipbe_ctx(&be_main)))) {
			fprintf(stderr, "psmc_boot: no device memory for the main run's tables (%s)\n", psmc_hip_strerror(rc));
			psmc_run_abort(main_run); main_run = 0;
		}
		if (!main_run) { be_main.destroy(be_main.self); bb.destroy(bb.self); psmc_options_free(&o); psmc_options_free(&om); return 1; }
	}
	const int status = psmc_boot_run(&o, n_rep, seed0, pattern, &bb, main_run);
	/* Every output file is written and closed.  Destroying the contexts of an exact job means handing 260 GB of tables back call by call: 2.0 s
	 * before the process may end; a process that just ends leaves that to the driver, which does it behind the next prompt (0.24 s to the
	 * exit; profiles/r06_boot_exit.txt -- the next process to take the SAME memory waits for the clearing either way).  PSMC_BOOT_TEARDOWN=1
	 * keeps the orderly way (leak checkers). */
	if (!getenv("PSMC_BOOT_TEARDOWN")) { fflush(0); _exit(status); }
	if (be_main.destroy) be_main.destroy(be_main.self);
	bb.destroy(bb.self);
	psmc_options_free(&o); psmc_options_free(&om);
	return status;
}
