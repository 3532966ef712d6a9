/* psmc_host.h -- host side of the MI355X PSMC driver: everything lh3/psmc does
 * around the E-step (command line, .psmcfa input, model -> HMM parameters,
 * Hooke-Jeeves M-step, .psmc output, bootstrap resampling, decoding output).
 * The E-step itself is NOT here: it is reached through `psmc_estep_backend`,
 * which the psmc binary binds to libpsmc_hip.so (include/psmc_hip.h).
 *
 * Written from scratch; evaluation order of every floating-point expression
 * follows the reference so that, given bit-identical sufficient statistics,
 * the .psmc output is byte-identical.  Citations are
 * file:line in the lh3/psmc checkout.
 */
#ifndef PSMC_HOST_H
#define PSMC_HOST_H
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PSMC_HOST_VERSION "0.6.5-r74-dirty" /* the MM Version line of the reference (cli.c:13), kept for byte parity */
#define PSMC_N_FIXED 3        /* theta0, rho0, max_t precede the lambdas (psmc.h:12) */
#define PSMC_T_INFINITY 1000.0 /* psmc.h:14 */

/* ---- -p pattern (cli.c:66-99) */
typedef struct {
	int n_states; /* number of atomic intervals = psmc's n + 1 */
	int n_free;   /* number of free lambda parameters */
	int *group;   /* group[k] in [0,n_free): which lambda interval k uses (par_map) */
} psmc_pattern;
int  psmc_pattern_parse(const char *text, psmc_pattern *out); /* 0 ok, -1 malformed */
void psmc_pattern_free(psmc_pattern *p);

/* ---- .psmcfa input (cli.c:15-32, 103-138; kseq.h:172-223) */
typedef struct {
	char *name;
	uint8_t *sym;    /* 0 hom, 1 het, 2 missing */
	int32_t L;       /* bins */
	int32_t L_called; /* bins that are not missing (L_e) */
	int32_t n_het;   /* het bins (n_e) */
} psmc_segment;
typedef struct {
	int n_seg;
	psmc_segment *seg;
	int64_t sum_called; /* sum of L_called: "sum_L" of the MM line */
	int64_t sum_het;    /* "sum_n" */
} psmc_input;
int  psmc_input_read(const char *path, psmc_input *in); /* path "-" = stdin; gz or plain */
void psmc_input_resample(psmc_input *in);                /* -b, aux.c:8-47 (drand48) */
int  psmc_input_resample_idx(const psmc_input *in, int32_t **idx); /* the same draw as a list of segment indices (caller frees); returns its length */
void psmc_input_free(psmc_input *in);
uint8_t psmc_symbol_of(unsigned char c);                 /* the 256-entry table of cli.c:15-32 */

/* ---- model: population parameters -> HMM parameters (core.c) */
typedef struct {
	psmc_pattern pat;
	char *pattern_text;
	double alpha;      /* -l, time-interval skew (cli.c:150) */
	int has_dt;        /* divergence model (-T) */
	double *fixed_t;   /* user time intervals from -i (inp_ti) or NULL */
	int n_params;
	double *params;    /* theta0, rho0, max_t, lambda[n_free], [dt] */
	/* derived by psmc_model_update */
	double *t;         /* n_states + 1 boundaries, t[n_states] = infinity */
	double *sigma, *post_sigma;
	double C_pi, C_sigma;
	double *a, *e, *a0; /* n*n, 3*n (row 2 = 1), n */
	/* EM bookkeeping printed every round */
	double lk, Q0, Q1;
	int fast_mstep;    /* 1: O(N) objective from the matrix factors (not bit-identical: fast mode only) */
} psmc_model;
psmc_model *psmc_model_new(const psmc_pattern *pat, const char *pattern_text, double alpha, int has_dt);
void psmc_model_free(psmc_model *m);
void psmc_model_update(psmc_model *m);                       /* psmc_update_hmm, core.c:61-133 */
void psmc_model_avg_t(const psmc_model *m, double *avg_t);   /* psmc_avg_t, core.c:135-162 */
int  psmc_model_logfactors(psmc_model *m, double *out7N);    /* logs of the factors of a[][] and e[][] (fast M-step) */
int  psmc_model_logfactors_simd(psmc_model *m, double *out7N); /* the same, vectorised (fastq.c; AVX2 + libmvec) */
void psmc_model_cap(psmc_model *m, int k0);                  /* psmc_cap_matrix, aux.c:115-127 */

/* ---- M-step pieces (khmm.c:326-382, kmin.c:48-107, em.c:15-25) */
double psmc_Q0(int n, const double *A, const double *E);
double psmc_Q(int n, const double *a, const double *e, const double *A, const double *E, double Q0);
typedef double (*psmc_objective)(int n, double *x, void *data);
double psmc_hooke_jeeves(psmc_objective f, int n, double *x, void *data, double r, double eps, int max_calls);

/* ---- the E-step seam */
typedef struct psmc_estep_backend {
	void *self;
	int  (*load)(void *self, int n_seg, const uint8_t *const *sym, const int32_t *L);
	/* em.c:33-55: A n*n, E 2*n (hom, het), LL; returns 0 or an error code */
	int  (*estep)(void *self, const double *a, const double *e, const double *a0, double *A, double *E, double *LL,
	              double *chk);
	/* hd->f, hd->b, hd->s of one segment, L*n / L*n / L (aux.c:157-158); any of the three may be NULL */
	int  (*tables)(void *self, int seg, double *f, double *b, double *s);
	/* optional (may be NULL): posterior argmax path[L] and its probability maxp[L] (khmm.c:264-281) */
	int  (*decode)(void *self, int seg, int32_t *path, double *maxp);
	/* optional (may be NULL): the E-step without the n*n counts -- sums = SL | SU | DG | CL | CU (5n: the triangular
	 * row / column sums of A and its diagonal), E 2n, LL; what the O(N) objective reads (fast M-step only) */
	int  (*estep_factored)(void *self, const double *a, const double *e, const double *a0, double *sums, double *E, double *LL);
	const char *(*error)(void *self);
	void (*destroy)(void *self);
	/* optional (may be NULL): full posterior post[L*n] and the DF line's recombination probability recomb[L] (aux.c:183-200) */
	int  (*posterior)(void *self, int seg, double *post, double *recomb);
	/* optional (may be NULL): cnt[n*n_cnt] += posterior-weighted counts of one segment's cntcpg record (aux.c:202-219) */
	int  (*post_counts)(void *self, int seg, const int32_t *cnt1, int32_t l, int32_t n_cnt, double *cnt);
} psmc_estep_backend;

/* ---- run state + driver (main.c, em.c, aux.c) */
typedef struct {
	/* options, defaults of cli.c:142-154 */
	int n_iters;      /* -N */
	double max_t;     /* -t */
	double tr_ratio;  /* -r */
	double alpha;     /* -l */
	double ran_init;  /* -I */
	double dt0;       /* -T (<0: off) */
	int cap_k;        /* -C */
	int decode, full_decode, print_prob, simulate, bootstrap; /* -d -D -s -S -b */
	int fast_mstep;   /* not a reference option: set by the driver for PSMC_HIP_MODE=fast (PSMC_FAST_MSTEP=0 to keep the exact objective) */
	char *pattern_text; /* -p */
	char *param_file; /* -i */
	char *cnt_file;   /* -c */
	char *out_file;   /* -o */
	char *in_file;
	FILE *out;
	/* not a reference option: an input the caller has read already (psmc's main() reads it on a thread while the device comes up);
	 * psmc_run_begin takes it over -- ownership included -- instead of reading in_file.  prefetch_rc: what psmc_input_read returned. */
	psmc_input *prefetched; int prefetch_rc;
} psmc_options;
void psmc_options_default(psmc_options *o);
int  psmc_options_parse(psmc_options *o, int argc, char **argv); /* 0 ok, 1 usage printed */
void psmc_options_free(psmc_options *o);

/* Whole program given an E-step backend: header, RD 0, n_iters EM rounds,
 * optional decoding / simulation.  Returns the process exit status. */
int psmc_run(psmc_options *o, psmc_estep_backend *be);
/* ... in two halves: everything up to RD 0 -- all draws from the process-wide drand48 stream happen here -- and the EM rounds,
 * decoding and output (frees the state).  psmc_run_begin returns NULL after printing what went wrong. */
typedef struct psmc_run_state psmc_run_state;
psmc_run_state *psmc_run_begin(psmc_options *o, psmc_estep_backend *be);
int psmc_run_finish(psmc_run_state *st);
void psmc_run_abort(psmc_run_state *st); /* begun, never to be finished: removes the output it opened, frees the state */

/* one EM round (psmc_em, em.c:27-78); prints the IT line to out */
int psmc_em_round(psmc_model *m, const psmc_input *in, psmc_estep_backend *be, FILE *out);
/* its M-step half (em.c:56-74) from given statistics: A n*n, or NULL with the five triangular sums `sums` (5n, fast
 * M-step only); E 2n; LL.  Prints IT, returns the number of objective calls.  Re-entrant per model. */
int psmc_em_mstep(psmc_model *m, const double *A, const double *E, const double *sums, double LL, FILE *out);

/* what comes before the first model: pattern or -i parameter file */
typedef struct { psmc_pattern pat; double *inp_pa, *inp_ti; int has_dt; } psmc_setup;
int  psmc_setup_begin(psmc_options *o, psmc_setup *su); /* 0 ok */
void psmc_setup_end(psmc_setup *su);
void psmc_print_header(const psmc_options *o, const psmc_pattern *pat, FILE *f); /* CC / MM lines, cli.c:200-224 */
psmc_model *psmc_model_start(const psmc_options *o, const psmc_setup *su, int64_t sum_called, int64_t sum_het); /* core.c:21-50 */

/* ---- bootstrap driver (config 4): n_rep replicates of `psmc -b` in one process.  Replicate r draws its trunks and
 * its initial parameters from srand48(seed0 + r) exactly as `PSMC_SEED=<seed0+r> psmc -b ...` does, and writes the
 * same .psmc stream to the file named by out_pattern (one %d = r).  Per EM iteration the E-steps of all replicates
 * go to the device(s) as one batch each, and a replicate's M-step runs on a host thread as soon as the batch reports its
 * statistics final -- under the rest of the batch. */
typedef struct psmc_batch_backend {
	void *self;
	int  n_dev; /* replicates are dealt round robin over the devices; one host thread drives each device */
	int  (*load)(void *self, int dev, int n_seg, const uint8_t *const *sym, const int32_t *L);
	/* n_rep E-steps over the loaded trunks: parameters a n_rep*n*n, e n_rep*2*n (rows hom, het), a0 n_rep*n;
	 * multisets sel_idx[sel_off[r] .. sel_off[r+1]); outputs A n_rep*n*n or NULL, sums n_rep*5n or NULL, E n_rep*2n, LL.
	 * `done(user, n, pos)` must be called (on the calling thread) for every replicate exactly once, as soon as its rows of the
	 * outputs are final -- at the latest before the call returns: the driver starts that replicate's M-step at once */
	int  (*estep_batch)(void *self, int dev, int n_rep, const double *a, const double *e, const double *a0,
	                    const int32_t *sel_off, const int32_t *sel_idx, double *A, double *sums, double *E, double *LL,
	                    void (*done)(void *user, int n_done, const int32_t *pos), void *user);
	const char *(*error)(void *self, int dev);
	void (*destroy)(void *self);
	int  can_factor; /* estep_batch can produce the triangular sums directly */
	/* optional (may be NULL): called once per device after the replicates are drawn and before the first E-step, with the table
	 * bins the device's replicates need together (the padded lengths of every replicate's UNIQUE trunks): the backend can take
	 * its table memory now, sized for exactly this job (psmc_hip_reserve_batch_tables) */
	int  (*reserve)(void *self, int dev, int64_t table_bins);
	/* optional (may be NULL): the main run that shared the first device (psmc_boot --main) has finished; called once per device, by
	 * the thread that drives it, between two of its batches: the backend can give the batch the whole device back */
	void (*main_done)(void *self, int dev);
} psmc_batch_backend;
/* main_run (may be NULL): a psmc_run_begin()'ed run -- the un-resampled main run of README:49-53 on its own input -- whose EM
 * rounds psmc_boot_run drives on a thread of its own beside the replicates (psmc_boot --main) */
int psmc_boot_run(psmc_options *o, int n_rep, long seed0, const char *out_pattern, psmc_batch_backend *bb, psmc_run_state *main_run);
void psmc_print_round(const psmc_model *m, int64_t sum_called, FILE *out); /* psmc_print_data, aux.c:49-82 */
int  psmc_usable_cpus(void); /* affinity mask capped by the control group's CPU quota (boot.c) */

/* synthetic data for benchmarks: hmm_simulate-like draw (khmm.c:386-423) with our own RNG */
void psmc_simulate_segment(int n, const double *a, const double *e, const double *a0, int32_t L, uint64_t seed,
                           double miss_rate, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
