/* boot.c -- config 4: the bootstrap farm of lh3/psmc (README:57-62 there: `seq 100 | xargs -i echo psmc -N25 ... -b
 * -o round-{}.psmc split.fa | sh`) as ONE process that keeps the trunks in HBM once and runs all replicates' EM
 * iterations together: the E-steps of a device's replicates go to it as one batch (psmc_hip_estep_batch_cb: exact
 * mode packs hundreds of trunk sweeps into one grid; fast mode keeps a learned tile plan per replicate), the
 * Hooke-Jeeves M-steps (host, em.c:56-68) run on host threads, one replicate each -- starting as soon as the batch
 * reports a replicate's statistics final, while the device is still busy with the rest of the batch.
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
	int it;                        /* the EM iteration A / E / sums / LL belong to */
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


/* ---- the EM iterations of all replicates as a two-stage pipeline: E-steps on the devices, M-steps on host threads.
 * A device thread sends ALL replicates of its device as one batch per EM iteration; the backend reports replicates as their statistics
 * become final (psmc_hip_estep_batch_cb: exact mode after each launch of the batch, fast mode after each replicate), and from that
 * moment a replicate's M-step (em.c:56-68) runs on one of the M-step threads -- under the rest of the batch.  A device starts its next
 * iteration when the M-steps of all ITS replicates are done; devices do not wait for each other. */
typedef struct {
	psmc_options *o; psmc_batch_backend *bb; replicate *rep; int n_rep, N, factored, timing;
	main_job *mj;
	pthread_mutex_t mu; pthread_cond_t cv; /* guard everything below */
	int *queue, q_head, q_tail;  /* ring of n_rep + 1 replicate numbers whose E-step results are in rep[] and whose M-step has not started */
	int *m_left;                 /* [device]: M-steps of the device's current iteration not finished yet */
	int stop, failed;
	/* per EM iteration, for PSMC_TIMING */
	int *it_done;                /* replicates through their M-step */
	double *e_ms, *e_end, *it_end, *m_work; /* longest batch call; when the last one ended; when the last M-step ended; summed M-step time */
} pipeline;
typedef struct { pipeline *P; int dev; pthread_t tid; } dev_job;
typedef struct { pipeline *P; int dev, it; const double *A, *S5, *E, *LL; } done_ctx;

static void pipe_fail(pipeline *P) { pthread_mutex_lock(&P->mu); P->failed = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); }

/* the backend's progress report: positions `pos` of the running batch are final -- copy them out, queue their M-steps */
static void batch_done(void *user, int n_done, const int32_t *pos)
{
	done_ctx *D = (done_ctx *)user;
	pipeline *P = D->P;
	const int N = P->N, nd = P->bb->n_dev;
	for (int i = 0; i < n_done; ++i) {
		const int j = pos[i];
		replicate *R = &P->rep[D->dev + j * nd];
		if (D->A) memcpy(R->A, D->A + (size_t)j * N * N, sizeof(double) * (size_t)N * N);
		if (D->S5) memcpy(R->sums, D->S5 + (size_t)j * 5 * N, sizeof(double) * (size_t)5 * N);
		memcpy(R->E, D->E + (size_t)j * 2 * N, sizeof(double) * (size_t)2 * N);
		R->LL = D->LL[j];
		R->it = D->it;
	}
	pthread_mutex_lock(&P->mu);
	for (int i = 0; i < n_done; ++i) { P->queue[P->q_tail] = D->dev + pos[i] * nd; P->q_tail = (P->q_tail + 1) % (P->n_rep + 1); }
	pthread_cond_broadcast(&P->cv);
	pthread_mutex_unlock(&P->mu);
}

static void *mstep_thread(void *arg)
{
	pipeline *P = (pipeline *)arg;
	for (;;) {
		pthread_mutex_lock(&P->mu);
		while (P->q_head == P->q_tail && !P->stop) pthread_cond_wait(&P->cv, &P->mu);
		if (P->q_head == P->q_tail) { pthread_mutex_unlock(&P->mu); return 0; } /* stop, nothing queued */
		const int r = P->queue[P->q_head];
		P->q_head = (P->q_head + 1) % (P->n_rep + 1);
		pthread_mutex_unlock(&P->mu);
		replicate *R = &P->rep[r];
		const double t0 = now_ms();
		/* independent models (em.c:56-74), then the round's output */
		psmc_em_mstep(R->m, R->A, R->E, R->sums, R->LL, R->out);
		fprintf(R->out, "RD\t%d\n", R->it + 1);
		psmc_print_round(R->m, R->sum_called, R->out);
		const double t1 = now_ms();
		pthread_mutex_lock(&P->mu);
		--P->m_left[r % P->bb->n_dev];
		++P->it_done[R->it];
		P->m_work[R->it] += t1 - t0;
		if (t1 > P->it_end[R->it]) P->it_end[R->it] = t1;
		pthread_cond_broadcast(&P->cv);
		pthread_mutex_unlock(&P->mu);
	}
}

/* one EM iteration's batch of device d: the E-steps of all its replicates; the results leave through batch_done */
static int estep_device(pipeline *P, int d, int it)
{
	psmc_batch_backend *bb = P->bb;
	replicate *rep = P->rep;
	const int N = P->N, nd = bb->n_dev;
	const int cnt = P->n_rep > d ? (P->n_rep - d + nd - 1) / nd : 0;
	int tot = 0;
	if (cnt == 0) return 0; /* more devices than replicates: nothing for this one */
	for (int j = 0; j < cnt; ++j) tot += rep[d + j * nd].n_idx;
	double *a = (double *)malloc(sizeof(double) * (size_t)cnt * N * N), *e = (double *)malloc(sizeof(double) * (size_t)cnt * 2 * N);
	double *a0 = (double *)malloc(sizeof(double) * (size_t)cnt * N);
	double *A = P->factored ? 0 : (double *)malloc(sizeof(double) * (size_t)cnt * N * N);
	double *S5 = P->factored ? (double *)malloc(sizeof(double) * (size_t)cnt * 5 * N) : 0;
	double *E = (double *)malloc(sizeof(double) * (size_t)cnt * 2 * N), *LL = (double *)malloc(sizeof(double) * (size_t)cnt);
	int32_t *off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cnt + 1)), *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
	off[0] = 0;
	for (int j = 0; j < cnt; ++j) {
		const replicate *R = &rep[d + j * nd];
		memcpy(a + (size_t)j * N * N, R->m->a, sizeof(double) * (size_t)N * N);
		memcpy(e + (size_t)j * 2 * N, R->m->e, sizeof(double) * (size_t)2 * N); /* rows hom, het; the missing row is implied */
		memcpy(a0 + (size_t)j * N, R->m->a0, sizeof(double) * (size_t)N);
		memcpy(idx + off[j], R->idx, sizeof(int32_t) * (size_t)R->n_idx);
		off[j + 1] = off[j] + R->n_idx;
	}
	done_ctx D = {P, d, it, A, S5, E, LL};
	pthread_mutex_lock(&P->mu);
	P->m_left[d] = cnt;
	pthread_mutex_unlock(&P->mu);
	const int rc = bb->estep_batch(bb->self, d, cnt, a, e, a0, off, idx, A, S5, E, LL, batch_done, &D);
	if (rc) fprintf(stderr, "psmc_boot: E-step batch failed on device %d: %s\n", d, bb->error(bb->self, d));
	free(a); free(e); free(a0); free(A); free(S5); free(E); free(LL); free(off); free(idx); /* (batch_done copied every row out) */
	return rc;
}

static void *dev_thread(void *arg)
{
	dev_job *J = (dev_job *)arg;
	pipeline *P = J->P;
	const int d = J->dev;
	int main_released = 0;
	for (int it = 0; it != P->o->n_iters; ++it) {
		if (P->mj && !main_released && P->bb->main_done && __atomic_load_n(&P->mj->finished, __ATOMIC_ACQUIRE)) {
			P->bb->main_done(P->bb->self, d); /* the main run is over: this device's batches get its compute units back */
			main_released = 1;
			if (P->timing) fprintf(stderr, "[psmc_boot] main run finished before iteration %d of device %d: its batches have the whole device again\n", it + 1, d);
		}
		const double t0 = now_ms();
		if (estep_device(P, d, it)) { pipe_fail(P); return 0; }
		const double t1 = now_ms();
		pthread_mutex_lock(&P->mu);
		if (t1 - t0 > P->e_ms[it]) P->e_ms[it] = t1 - t0;
		if (t1 > P->e_end[it]) P->e_end[it] = t1;
		while (!P->failed && P->m_left[d] > 0) pthread_cond_wait(&P->cv, &P->mu); /* the parameters of the next iteration exist when every M-step of this one is through */
		const int ok = !P->failed;
		pthread_mutex_unlock(&P->mu);
		if (!ok) return 0;
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
	/* (every way out before the main run's thread exists takes the begun main run with it: psmc_run_abort -- ADVICE r5) */
	if (n_rep < 1 || !pat_ok || !pat_d) { fprintf(stderr, "psmc_boot: need a replicate count and an output pattern with exactly one %%d (and no other conversion)\n"); psmc_run_abort(main_run); return 1; }
	if (o->decode || o->cnt_file || o->print_prob || o->simulate) { fprintf(stderr, "psmc_boot: decoding / simulation options make no sense on bootstrap replicates\n"); psmc_run_abort(main_run); return 1; }
	if (psmc_setup_begin(o, &su)) { psmc_run_abort(main_run); return 1; }
	const int N = su.pat.n_states;
	if (psmc_input_read(o->in_file, &in) || in.n_seg == 0) { fprintf(stderr, "psmc_boot: no sequence in %s\n", o->in_file); psmc_setup_end(&su); psmc_run_abort(main_run); return 1; }
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
	if (bb->reserve) { /* the replicates are drawn: every device learns what its batch will need, and takes it before the first E-step (250 GB of an exact job's tables
	                     * are 6 s on a device whose memory was in use before -- the driver clears what it hands out; starting the main run's thread
	                     * first does not hide them: the runtime serialises its calls behind the allocation, profiles/r06_boot_exit.txt) */
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
	{ /* main.c:16-20 for every replicate */
		pipeline P;
		memset(&P, 0, sizeof P);
		P.o = o; P.bb = bb; P.rep = rep; P.n_rep = n_rep; P.N = N; P.factored = factored; P.mj = main_started ? &mj : 0; P.timing = timing;
		P.failed = failed;
		pthread_mutex_init(&P.mu, 0); pthread_cond_init(&P.cv, 0);
		const size_t ni = (size_t)(o->n_iters > 0 ? o->n_iters : 1);
		P.queue = (int *)calloc((size_t)n_rep + 1, sizeof(int));
		P.m_left = (int *)calloc((size_t)bb->n_dev, sizeof(int));
		P.it_done = (int *)calloc(ni, sizeof(int));
		P.e_ms = (double *)calloc(ni, sizeof(double)); P.e_end = (double *)calloc(ni, sizeof(double));
		P.it_end = (double *)calloc(ni, sizeof(double)); P.m_work = (double *)calloc(ni, sizeof(double));
		/* M-step threads: what the process may use, less the threads that drive the devices (they spin inside the HIP runtime while a batch
		 * runs), the main run's, and two for the runtime's own helpers -- the M-steps now run WHILE the device threads feed the device, and a
		 * process that overdraws its CPU quota is stopped whole; OMP_NUM_THREADS, when set, is taken as given (the M-steps were an OpenMP
		 * loop until round 5) */
		int m_threads = psmc_usable_cpus() - bb->n_dev - (main_started ? 1 : 0) - 2;
		if (m_threads < 1) m_threads = 1;
		if (getenv("OMP_NUM_THREADS") && atoi(getenv("OMP_NUM_THREADS")) > 0) m_threads = atoi(getenv("OMP_NUM_THREADS"));
		if (m_threads > n_rep) m_threads = n_rep;
		if (timing) fprintf(stderr, "[psmc_boot] %d usable processors, %d M-step threads, %d device thread(s)%s\n", psmc_usable_cpus(), m_threads, bb->n_dev, main_started ? ", 1 main-run thread" : "");
		pthread_t *mt = (pthread_t *)calloc((size_t)m_threads, sizeof(pthread_t));
		dev_job *dj = (dev_job *)calloc((size_t)bb->n_dev, sizeof(dev_job));
		int n_mt = 0, n_thr = 0;
		for (int k = 0; k < m_threads && !P.failed; ++k) {
			if (pthread_create(&mt[k], 0, mstep_thread, &P) == 0) ++n_mt;
			else if (n_mt == 0) { fprintf(stderr, "psmc_boot: cannot start an M-step thread\n"); pipe_fail(&P); }
			else break; /* fewer threads than asked for */
		}
		for (int d = 0; d < bb->n_dev && !P.failed; ++d) {
			dj[d].P = &P; dj[d].dev = d;
			if (pthread_create(&dj[d].tid, 0, dev_thread, &dj[d]) == 0) ++n_thr;
			else { fprintf(stderr, "psmc_boot: cannot start the thread of device %d\n", d); pipe_fail(&P); }
		}
		if (timing) { /* one line per EM iteration, when its last M-step is through */
			double t_prev = now_ms();
			for (int it = 0; it != o->n_iters; ++it) {
				pthread_mutex_lock(&P.mu);
				while (!P.failed && P.it_done[it] < n_rep) pthread_cond_wait(&P.cv, &P.mu);
				const int ok = !P.failed;
				const double e_ms = P.e_ms[it], tail = P.it_end[it] - P.e_end[it], work = P.m_work[it], t_end = P.it_end[it];
				pthread_mutex_unlock(&P.mu);
				if (!ok) break;
				fprintf(stderr, "[psmc_boot] iteration %d: %d E-steps %.1f ms on %d device(s), M-steps %.1f ms after the last batch (%.0f ms of work on %d threads, the rest under the batches), wall %.1f ms\n",
				        it + 1, n_rep, e_ms, bb->n_dev, tail > 0.0 ? tail : 0.0, work, n_mt, t_end - t_prev);
				t_prev = t_end;
			}
		}
		for (int d = 0; d < n_thr; ++d) pthread_join(dj[d].tid, 0);
		pthread_mutex_lock(&P.mu);
		P.stop = 1;
		pthread_cond_broadcast(&P.cv);
		pthread_mutex_unlock(&P.mu);
		for (int k = 0; k < n_mt; ++k) pthread_join(mt[k], 0);
		failed = P.failed;
		free(mt); free(dj); free(P.queue); free(P.m_left); free(P.it_done); free(P.e_ms); free(P.e_end); free(P.it_end); free(P.m_work);
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
	if (main_run) { psmc_run_abort(main_run); main_run = 0; } /* begun, its thread never started */
	psmc_input_free(&in);
	psmc_setup_end(&su);
	return status;
}
