/* boot.c -- config 4: the bootstrap farm of lh3/psmc (README:57-62 there: `seq 100 | xargs -i echo psmc -N25 ... -b
 * -o round-{}.psmc split.fa | sh`) as ONE process that keeps the trunks in HBM once and runs all replicates' EM
 * iterations together: the E-steps of a group of replicates go to a device as one batch (psmc_hip_estep_batch: exact
 * mode packs hundreds of trunk sweeps into one grid; fast mode keeps a learned tile plan per replicate), the
 * Hooke-Jeeves M-steps (host, em.c:56-68) run on host threads, one replicate each -- and while they run, the device
 * is already busy with the E-steps of the device's other group (two groups per device: PSMC_BOOT_GROUPS).
 *
 * Replicate r is `PSMC_SEED=<seed0+r> psmc -b <options>`: the same srand48 seed, hence the same psmc_resamp draw
 * (aux.c:8-47) and the same -I initial parameters, the same .psmc stream -- byte for byte in exact mode
 * (tests/test_host_cli.py).  Replicates are independent whole EM runs: no collective; with several devices they are
 * dealt round robin, one host thread driving each device.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "psmc_host.h"

/* The processors this process may really use: its affinity mask, capped by the CPU quota of its control group (v2 cpu.max, v1
 * cpu.cfs_quota_us).  A container on a 256-thread host with a quota of 16 shows 256 processors to OpenMP: 256 M-step threads spend
 * the quota of a 100 ms scheduler period in its first milliseconds, and the kernel then stops EVERY thread of the group -- the ones
 * that feed the device included -- until the period ends (measured: iterations quantised to multiples of 100 ms). */
int psmc_usable_cpus(void)
{
	int n = 0;
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
	if (n < 1) n = 1;
	double quota = -1.0, period = 0.0;
	FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (fp) {
		char q[64];
		if (fscanf(fp, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
		fclose(fp);
	} else {
		FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fr = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
		if (fq && fr && fscanf(fq, "%lf", &quota) == 1 && fscanf(fr, "%lf", &period) == 1) { /* (-1: no quota) */ }
		else quota = -1.0;
		if (fq) fclose(fq);
		if (fr) fclose(fr);
	}
	if (quota > 0.0 && period > 0.0) {
		const int c = (int)(quota / period + 0.5);
		if (c >= 1 && c < n) n = c;
	}
	return n;
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

typedef struct {
	psmc_model *m;
	int32_t *idx; int n_idx;       /* the drawn trunks, in order, repeats allowed */
	int64_t sum_called, sum_het;
	FILE *out;
	double *A, *E, *sums, LL;
} replicate;

/* the main run's EM rounds (psmc_run_finish) on a thread of their own */
typedef struct { psmc_run_state *st; int status; double ms; volatile int finished; } main_job;
static void *main_thread(void *arg)
{
	main_job *j = (main_job *)arg;
	const double t0 = now_ms();
	j->status = psmc_run_finish(j->st);
	j->ms = now_ms() - t0;
	__atomic_store_n(&j->finished, 1, __ATOMIC_RELEASE);
	return 0;
}


/* ---- the EM iterations of all replicates as a two-stage pipeline: E-steps on the devices, M-steps on host threads */
#define MAX_GRP 8
typedef struct {
	psmc_options *o; psmc_batch_backend *bb; replicate *rep; int n_rep, N, factored, n_grp, timing;
	main_job *mj;
	pthread_mutex_t mu; pthread_cond_t cv; /* guard everything below */
	int *e_done;        /* [device * n_grp + group]: EM iterations whose E-step results of that group are in rep[] */
	int m_done[MAX_GRP]; /* EM iterations whose M-steps of the group (all devices) are done */
	double *e_ms;       /* [device * n_iters + iteration]: time inside estep_batch */
	int failed;
} pipeline;
typedef struct { pipeline *P; int dev; pthread_t tid; } dev_job;

static void pipe_fail(pipeline *P) { pthread_mutex_lock(&P->mu); P->failed = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); }
static int pipe_failed(pipeline *P) { pthread_mutex_lock(&P->mu); const int f = P->failed; pthread_mutex_unlock(&P->mu); return f; }
/* device d drives replicates d, d + n_dev, ...: positions [lo, hi) of that list are its group g.  In the FIRST EM iteration group 0 is
 * the whole list and the others are empty: the backend's first batch call sees every replicate of the device, so what it derives
 * from that call (fast mode: ONE tile length for all replicates, from the largest) does not depend on the grouping */
static void group_range(const pipeline *P, int d, int g, int it, int *lo, int *hi)
{
	const int cnt = P->n_rep > d ? (P->n_rep - d + P->bb->n_dev - 1) / P->bb->n_dev : 0;
	if (it == 0) { *lo = 0; *hi = g == 0 ? cnt : 0; return; }
	*lo = (int)((int64_t)cnt * g / P->n_grp); *hi = (int)((int64_t)cnt * (g + 1) / P->n_grp);
}

/* one batch: the E-steps of positions [lo, hi) of device d's replicates, results into rep[] */
static int estep_group(pipeline *P, int d, int lo, int hi)
{
	psmc_batch_backend *bb = P->bb;
	replicate *rep = P->rep;
	const int N = P->N, cnt = hi - lo, nd = bb->n_dev;
	int tot = 0;
	for (int j = lo; j < hi; ++j) tot += rep[d + j * nd].n_idx;
	double *a = (double *)malloc(sizeof(double) * (size_t)cnt * N * N), *e = (double *)malloc(sizeof(double) * (size_t)cnt * 2 * N);
	double *a0 = (double *)malloc(sizeof(double) * (size_t)cnt * N);
	double *A = P->factored ? 0 : (double *)malloc(sizeof(double) * (size_t)cnt * N * N);
	double *S5 = P->factored ? (double *)malloc(sizeof(double) * (size_t)cnt * 5 * N) : 0;
	double *E = (double *)malloc(sizeof(double) * (size_t)cnt * 2 * N), *LL = (double *)malloc(sizeof(double) * (size_t)cnt);
	int32_t *off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cnt + 1)), *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
	off[0] = 0;
	for (int j = 0; j < cnt; ++j) {
		const replicate *R = &rep[d + (lo + j) * nd];
		memcpy(a + (size_t)j * N * N, R->m->a, sizeof(double) * (size_t)N * N);
		memcpy(e + (size_t)j * 2 * N, R->m->e, sizeof(double) * (size_t)2 * N); /* rows hom, het; the missing row is implied */
		memcpy(a0 + (size_t)j * N, R->m->a0, sizeof(double) * (size_t)N);
		memcpy(idx + off[j], R->idx, sizeof(int32_t) * (size_t)R->n_idx);
		off[j + 1] = off[j] + R->n_idx;
	}
	const int rc = bb->estep_batch(bb->self, d, lo, cnt, a, e, a0, off, idx, A, S5, E, LL);
	if (rc) fprintf(stderr, "psmc_boot: E-step batch failed on device %d: %s\n", d, bb->error(bb->self, d));
	for (int j = 0; j < cnt && !rc; ++j) {
		replicate *R = &rep[d + (lo + j) * nd];
		if (A) memcpy(R->A, A + (size_t)j * N * N, sizeof(double) * (size_t)N * N);
		if (S5) memcpy(R->sums, S5 + (size_t)j * 5 * N, sizeof(double) * (size_t)5 * N);
		memcpy(R->E, E + (size_t)j * 2 * N, sizeof(double) * (size_t)2 * N);
		R->LL = LL[j];
	}
	free(a); free(e); free(a0); free(A); free(S5); free(E); free(LL); free(off); free(idx);
	return rc;
}

static void *dev_thread(void *arg)
{
	dev_job *J = (dev_job *)arg;
	pipeline *P = J->P;
	const int d = J->dev;
	int main_released = 0;
	for (int it = 0; it != P->o->n_iters; ++it)
		for (int g = 0; g < P->n_grp; ++g) {
			int lo, hi, ok;
			pthread_mutex_lock(&P->mu);
			while (!P->failed && P->m_done[g] < it) pthread_cond_wait(&P->cv, &P->mu); /* the group's parameters of this iteration exist */
			ok = !P->failed;
			pthread_mutex_unlock(&P->mu);
			if (!ok) return 0;
			if (P->mj && !main_released && P->bb->main_done && __atomic_load_n(&P->mj->finished, __ATOMIC_ACQUIRE)) {
				P->bb->main_done(P->bb->self, d); /* the main run is over: this device's batches get its compute units back */
				main_released = 1;
				if (P->timing) fprintf(stderr, "[psmc_boot] main run finished before iteration %d, group %d of device %d: its batches have the whole device again\n", it + 1, g, d);
			}
			group_range(P, d, g, it, &lo, &hi);
			const double t0 = now_ms();
			if (hi > lo && estep_group(P, d, lo, hi)) { pipe_fail(P); return 0; }
			pthread_mutex_lock(&P->mu);
			P->e_ms[(size_t)d * P->o->n_iters + it] += now_ms() - t0;
			P->e_done[d * P->n_grp + g] = it + 1;
			pthread_cond_broadcast(&P->cv);
			pthread_mutex_unlock(&P->mu);
		}
	return 0;
}

int psmc_boot_run(psmc_options *o, int n_rep, long seed0, const char *out_pattern, psmc_batch_backend *bb, psmc_run_state *main_run)
{
	psmc_setup su;
	psmc_input in;
	int status = 1;
	/* the pattern is the user's string: never a printf format.  Exactly one "%d" (the replicate number); "%%" = a literal % */
	const char *pat_d = 0;
	int pat_ok = out_pattern != 0;
	for (const char *q = out_pattern; pat_ok && *q; ++q) {
		if (*q != '%') continue;
		if (q[1] == '%') { ++q; continue; }
		if (q[1] == 'd' && !pat_d) { pat_d = q; ++q; continue; }
		pat_ok = 0; /* a second %d, or any other conversion */
	}
	if (pat_ok && strlen(out_pattern) + 16 >= 4096) pat_ok = 0; /* the expansion below writes into char fn[4096]: a longer pattern would be cut, and every replicate would open the same file */
	if (n_rep < 1 || !pat_ok || !pat_d) { fprintf(stderr, "psmc_boot: need a replicate count and an output pattern with exactly one %%d (and no other conversion)\n"); return 1; }
	if (o->decode || o->cnt_file || o->print_prob || o->simulate) { fprintf(stderr, "psmc_boot: decoding / simulation options make no sense on bootstrap replicates\n"); return 1; }
	if (psmc_setup_begin(o, &su)) return 1;
	const int N = su.pat.n_states;
	if (psmc_input_read(o->in_file, &in) || in.n_seg == 0) { fprintf(stderr, "psmc_boot: no sequence in %s\n", o->in_file); psmc_setup_end(&su); return 1; }
	for (int i = 0; i < in.n_seg; ++i)
		if (in.seg[i].L < 1) { fprintf(stderr, "psmc_boot: empty sequence '%s'\n", in.seg[i].name); goto done_input; }
	{ /* the trunks go to every device once */
		const size_t ns = (size_t)(in.n_seg > 0 ? in.n_seg : 1);
		const uint8_t **ptr = (const uint8_t **)malloc(sizeof(void *) * ns);
		int32_t *len = (int32_t *)malloc(sizeof(int32_t) * ns);
		for (int i = 0; i < in.n_seg; ++i) { ptr[i] = in.seg[i].sym; len[i] = in.seg[i].L; }
		int rc = 0;
		for (int d = 0; d < bb->n_dev && rc == 0; ++d) {
			rc = bb->load(bb->self, d, in.n_seg, ptr, len);
			if (rc) fprintf(stderr, "psmc_boot: cannot load the trunks on device %d: %s\n", d, bb->error(bb->self, d));
		}
		free(ptr); free(len);
		if (rc) goto done_input;
	}
	replicate *rep = (replicate *)calloc((size_t)n_rep, sizeof(replicate));
	const int factored = o->fast_mstep && bb->can_factor && N <= 128;
	for (int r = 0; r < n_rep; ++r) { /* serial: drand48 is one global stream, re-seeded per replicate like a fresh process */
		replicate *R = &rep[r];
		char fn[4096];
		{ /* prefix + r + suffix, "%%" -> "%" */
			size_t w = 0;
			for (const char *q = out_pattern; *q && w + 16 < sizeof fn; ++q) {
				if (q == pat_d) { w += (size_t)snprintf(fn + w, sizeof fn - w, "%d", r); ++q; }
				else if (*q == '%') { fn[w++] = '%'; ++q; }
				else fn[w++] = *q;
			}
			fn[w] = 0;
		}
		R->out = fopen(fn, "w");
		if (!R->out) { fprintf(stderr, "psmc_boot: cannot write %s\n", fn); goto done_rep; }
		srand48(seed0 + r);                                     /* main.c:11 with PSMC_SEED */
		psmc_print_header(o, &su.pat, R->out);
		R->n_idx = psmc_input_resample_idx(&in, &R->idx);       /* -b: aux.c:8-47 */
		for (int i = 0; i < R->n_idx; ++i) { R->sum_called += in.seg[R->idx[i]].L_called; R->sum_het += in.seg[R->idx[i]].n_het; }
		fprintf(R->out, "MM\tn_seqs:%d, sum_L:%lld, sum_n:%d\n", R->n_idx, (long long)R->sum_called, (int)R->sum_het);
		R->m = psmc_model_start(o, &su, R->sum_called, R->sum_het);
		fprintf(R->out, "RD\t0\n");
		psmc_print_round(R->m, R->sum_called, R->out);
		R->A = factored ? 0 : (double *)calloc((size_t)N * N, sizeof(double));
		R->sums = factored ? (double *)calloc((size_t)5 * N, sizeof(double)) : 0;
		R->E = (double *)calloc((size_t)2 * N, sizeof(double));
	}
	if (bb->reserve) { /* the replicates are drawn: every device learns what its batch will need, and takes it before the first E-step */
		for (int d = 0; d < bb->n_dev; ++d) {
			int64_t bins = 0;
			unsigned char *seen = (unsigned char *)malloc((size_t)(in.n_seg > 0 ? in.n_seg : 1));
			for (int r = d; r < n_rep; r += bb->n_dev) {
				memset(seen, 0, (size_t)in.n_seg);
				for (int i = 0; i < rep[r].n_idx; ++i) {
					const int32_t sg = rep[r].idx[i];
					if (!seen[sg]) { seen[sg] = 1; bins += ((int64_t)in.seg[sg].L + 63) & ~(int64_t)63; }
				}
			}
			free(seen);
			const int rc = bb->reserve(bb->self, d, bins);
			if (rc) { fprintf(stderr, "psmc_boot: cannot reserve the tables on device %d: %s\n", d, bb->error(bb->self, d)); goto done_rep; }
		}
	}
	const int timing = getenv("PSMC_TIMING") != 0;
	int failed = 0;
	/* The main run starts now: every draw from drand48 -- its own (psmc_run_begin) and the replicates' (above) -- is done, and from here
	 * on the two only share the device: the main run's sweeps on the compute units its context was given, the batch on the others. */
	main_job mj = {main_run, 0, 0.0, 0};
	pthread_t main_tid;
	int main_started = 0;
	if (main_run) {
		if (pthread_create(&main_tid, 0, main_thread, &mj) == 0) main_started = 1;
		else { fprintf(stderr, "psmc_boot: cannot start the main run's thread\n"); failed = 1; }
	}
	{ /* main.c:16-20 for every replicate.  Each device's replicates form n_grp groups; a device thread sends group after group to
	   * the device, and the M-steps of a group (host threads, here) run while the device is busy with the NEXT group's E-steps:
	   * E(all,1) | M(all,1) | E(g0,2) | E(g1,2) + M(g0,2) | E(g0,3) + M(g1,2) | ...  A replicate still sees E, M, E, M, ... in order: same output. */
		pipeline P;
		const char *gs = getenv("PSMC_BOOT_GROUPS");
		memset(&P, 0, sizeof P);
		P.n_grp = gs ? atoi(gs) : 2;
		if (P.n_grp < 1) P.n_grp = 1;
		if (P.n_grp > MAX_GRP) P.n_grp = MAX_GRP;
		if ((n_rep + bb->n_dev - 1) / bb->n_dev < P.n_grp) P.n_grp = 1; /* (no device with a replicate for every group: one batch per iteration) */
		P.o = o; P.bb = bb; P.rep = rep; P.n_rep = n_rep; P.N = N; P.factored = factored; P.mj = main_started ? &mj : 0; P.timing = timing;
		P.failed = failed;
		pthread_mutex_init(&P.mu, 0); pthread_cond_init(&P.cv, 0);
		P.e_done = (int *)calloc((size_t)bb->n_dev * P.n_grp, sizeof(int));
		P.e_ms = (double *)calloc((size_t)bb->n_dev * (size_t)(o->n_iters > 0 ? o->n_iters : 1), sizeof(double));
		dev_job *dj = (dev_job *)calloc((size_t)bb->n_dev, sizeof(dev_job));
		int n_thr = 0;
		for (int d = 0; d < bb->n_dev && !P.failed; ++d) {
			dj[d].P = &P; dj[d].dev = d;
			if (pthread_create(&dj[d].tid, 0, dev_thread, &dj[d]) == 0) ++n_thr;
			else { fprintf(stderr, "psmc_boot: cannot start the thread of device %d\n", d); pipe_fail(&P); }
		}
		int *list = (int *)malloc(sizeof(int) * (size_t)n_rep);
		/* M-step threads: what the process may use, less the threads that drive the devices (they spin inside the HIP runtime while a batch
		 * runs) and the main run's; OMP_NUM_THREADS, when set, is taken as given */
		int m_threads = psmc_usable_cpus() - bb->n_dev - (main_started ? 1 : 0);
		if (m_threads < 1) m_threads = 1;
		if (getenv("OMP_NUM_THREADS") && atoi(getenv("OMP_NUM_THREADS")) > 0) m_threads = atoi(getenv("OMP_NUM_THREADS"));
		if (timing) fprintf(stderr, "[psmc_boot] %d usable processors, %d M-step threads, %d device thread(s)%s\n", psmc_usable_cpus(), m_threads, bb->n_dev, main_started ? ", 1 main-run thread" : "");
		double t_prev = now_ms();
		for (int it = 0; it != o->n_iters && !pipe_failed(&P); ++it) {
			double m_ms = 0.0;
			for (int g = 0; g < P.n_grp; ++g) {
				int n_list = 0, ok = 1;
				pthread_mutex_lock(&P.mu);
				for (int d = 0; d < bb->n_dev; ++d)
					while (!P.failed && P.e_done[d * P.n_grp + g] < it + 1) pthread_cond_wait(&P.cv, &P.mu);
				ok = !P.failed;
				pthread_mutex_unlock(&P.mu);
				if (!ok) break;
				for (int d = 0; d < bb->n_dev; ++d) {
					int lo, hi; group_range(&P, d, g, it, &lo, &hi);
					for (int j = lo; j < hi; ++j) list[n_list++] = d + j * bb->n_dev;
				}
				const double t1 = now_ms();
				/* M-steps: independent models, one host thread each (em.c:56-74), then the round's output */
#pragma omp parallel for schedule(dynamic, 1) num_threads(m_threads)
				for (int q = 0; q < n_list; ++q) {
					replicate *R = &rep[list[q]];
					psmc_em_mstep(R->m, R->A, R->E, R->sums, R->LL, R->out);
					fprintf(R->out, "RD\t%d\n", it + 1);
					psmc_print_round(R->m, R->sum_called, R->out);
				}
				m_ms += now_ms() - t1;
				pthread_mutex_lock(&P.mu);
				P.m_done[g] = it + 1;
				pthread_cond_broadcast(&P.cv);
				pthread_mutex_unlock(&P.mu);
			}
			if (timing && !pipe_failed(&P)) {
				double e_ms = 0.0;
				pthread_mutex_lock(&P.mu);
				for (int d = 0; d < bb->n_dev; ++d) if (P.e_ms[(size_t)d * o->n_iters + it] > e_ms) e_ms = P.e_ms[(size_t)d * o->n_iters + it];
				pthread_mutex_unlock(&P.mu);
				const double t_now = now_ms();
				fprintf(stderr, "[psmc_boot] iteration %d: %d E-steps %.1f ms on %d device(s), M-steps %.1f ms, %d group(s), wall %.1f ms\n", it + 1, n_rep, e_ms, bb->n_dev, m_ms, P.n_grp, t_now - t_prev);
				t_prev = t_now;
			}
		}
		for (int d = 0; d < n_thr; ++d) pthread_join(dj[d].tid, 0);
		failed = P.failed;
		free(list); free(dj); free(P.e_done); free(P.e_ms);
		pthread_mutex_destroy(&P.mu); pthread_cond_destroy(&P.cv);
	}
	status = failed ? 1 : 0;
	if (main_started) {
		pthread_join(main_tid, 0);
		main_run = 0; /* psmc_run_finish freed it */
		if (mj.status) { fprintf(stderr, "psmc_boot: the main run failed\n"); status = 1; }
		if (timing) fprintf(stderr, "[psmc_boot] main run: %d EM iterations beside the replicates in %.1f ms\n", o->n_iters, mj.ms);
	}
done_rep:
	for (int r = 0; r < n_rep; ++r) {
		if (rep[r].out) fclose(rep[r].out);
		if (rep[r].m) psmc_model_free(rep[r].m);
		free(rep[r].idx); free(rep[r].A); free(rep[r].E); free(rep[r].sums);
	}
	free(rep);
done_input:
	psmc_input_free(&in);
	psmc_setup_end(&su);
	return status;
}
