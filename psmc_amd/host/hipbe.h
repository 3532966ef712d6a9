/* hipbe.h -- the E-step backend of the host driver bound to libpsmc_hip.so (include/psmc_hip.h): what `psmc` and the main
 * run of `psmc_boot --main` hand to psmc_run().  Compiled into the executables, not into libpsmc_host.so (which stays
 * free of GPU code: the CPU tests link it against the oracle instead). */
#ifndef PSMC_HIPBE_H
#define PSMC_HIPBE_H
#include "psmc_host.h"
#include "psmc_hip.h"

/* One context on `device` (devices == NULL or without a comma) or the segments of every E-step sharded over the listed
 * devices (psmc_hip_group_*).  use_factored: hand psmc_run the factored E-step (fast mode + O(N) objective).
 * Returns 0 or a PSMC_HIP_E* code; be->destroy releases everything. */
int psmc_hipbe_create(psmc_estep_backend *be, int n_states, int mode, int use_factored, const char *devices, int device);
/* the single context behind the backend (NULL for a sharded one): psmc_hip_set_cu_range, psmc_hip_reserve_tables */
psmc_hip_ctx *psmc_hipbe_ctx(psmc_estep_backend *be);
#endif
