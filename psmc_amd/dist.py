"""Sharding the E-step over the GPUs of one node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

Segments are independent given the HMM parameters (lh3/psmc em.c:36-55: the
only state carried across the segment loop is `he_sum += he` and `LL +=`), so
each rank runs the E-step of its own segments and ONE collective per EM
iteration replaces hmm_add_expect (khmm.c:346-359):

  fast mode   all_reduce(SUM) of the n*n + 2n + 1 doubles [A | E | LL] straight
              from the device buffer the reduction kernel wrote (34 KB at n=64:
              latency-bound on xGMI, any algorithm);
  exact mode  all_gather of the per-segment `he` of every rank, then every rank
              adds them in the GLOBAL input order -- floating-point addition is
              not associative, and bit-identity with the reference's serial
              loop is the point of this mode.

The longest-processing-time partition keeps the per-rank sum of segment lengths
balanced (the exact mode's critical path is its longest segment anyway).
"""
import numpy as np


def partition_segments(lengths, world):
    """Longest-processing-time-first: returns `world` sorted index lists covering range(len(lengths))."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    parts = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += int(lengths[i])
    return [sorted(p) for p in parts]


class ShardedEStep:
    """E-step of `segs` spread over the ranks of the default process group.

    make_engine(n_states, local_segments, mode) -> object with
        .estep(a, e, a0) -> dict(A, E, LL)                       (fast mode)
        .estep_segments(a, e, a0) -> dict(seg_A, seg_E, seg_LL)  (exact mode)
    The default engine is the HIP library on this rank's GPU.
    """

    def __init__(self, n_states, segs, mode="fast", make_engine=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n = int(n_states)
        self.mode = mode
        self.n_seg = len(segs)
        lens = [len(s) for s in segs]
        self.parts = partition_segments(lens, self.world)
        self.mine = self.parts[self.rank]
        local = [segs[i] for i in self.mine]
        self.device = None      # set when the engine is the HIP library: this rank's GPU
        self._stats = None      # device-resident [A | E | LL] the collective runs on (HIP engine, fast mode)
        if make_engine is None:
            import os
            import torch
            from . import hip
            if device is None:  # one process per GPU: torchrun's LOCAL_RANK, else round-robin over the visible devices
                nd = max(1, torch.cuda.device_count())
                device = int(os.environ.get("LOCAL_RANK", self.rank % nd)) % nd
            self.device = int(device)
            torch.cuda.set_device(self.device)

            def make_engine(n, local_segs, mode_):
                es = hip.HipEStep(n, device=self.device, mode=hip.MODE_FAST if mode_ == "fast" else hip.MODE_EXACT)
                if local_segs:
                    es.load_segments(local_segs)
                return es
        self.engine = make_engine(self.n, local, mode) if local else None

    def _estep_fast_device(self, a, e, a0):
        """HIP engine: the E-step leaves [A | E | LL] in HBM and the collective reduces that buffer in place (RCCL over
        xGMI with the nccl backend); nothing but the 34 KB result crosses PCIe, once, after the all-reduce."""
        import torch
        n = self.n
        dev = torch.device("cuda", self.device)
        if self._stats is None:
            self._stats = torch.zeros(n * n + 2 * n + 1, dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev)
        if self.engine is not None:
            self.engine.estep_device(a, e, a0, self._stats.data_ptr(), st.cuda_stream)
        else:
            self._stats.zero_()
        if self.world > 1:
            if self.dist.get_backend() == "nccl":
                self.dist.all_reduce(self._stats)  # replaces hmm_add_expect across shards (khmm.c:346-359)
            else:                                  # gloo (tests on one GPU / CPU rendezvous): reduce a host copy
                st.synchronize()
                t = self._stats.cpu()
                self.dist.all_reduce(t)
                self._stats.copy_(t)
        st.synchronize()
        return self._stats.cpu().numpy()

    def estep(self, a, e, a0):
        import torch
        n = self.n
        if self.mode == "fast":
            if self.device is not None:
                v = self._estep_fast_device(a, e, a0)
            else:  # injected engine (CPU tests): host vectors
                vec = np.zeros(n * n + 2 * n + 1)
                if self.engine is not None:
                    r = self.engine.estep(a, e, a0)
                    vec = np.concatenate([r["A"].ravel(), r["E"].ravel(), [r["LL"]]])
                t = torch.from_numpy(vec)
                if self.world > 1:
                    self.dist.all_reduce(t)  # replaces hmm_add_expect across shards
                v = t.numpy()
            return dict(A=v[:n * n].reshape(n, n).copy(), E=v[n * n:n * n + 2 * n].reshape(2, n).copy(), LL=float(v[-1]))
        # exact: gather every segment's own statistics, add in input order on every rank
        mine = dict(idx=self.mine, seg_A=np.zeros((0, n, n)), seg_E=np.zeros((0, 3, n)), seg_LL=np.zeros(0))
        if self.engine is not None:
            r = self.engine.estep_segments(a, e, a0)
            mine.update(seg_A=r["seg_A"], seg_E=r["seg_E"], seg_LL=r["seg_LL"])
        if self.world > 1:
            gathered = [None] * self.world
            self.dist.all_gather_object(gathered, mine)
        else:
            gathered = [mine]
        segA = [None] * self.n_seg; segE = [None] * self.n_seg; segLL = [None] * self.n_seg
        for g in gathered:
            for j, i in enumerate(g["idx"]):
                segA[i] = g["seg_A"][j]; segE[i] = g["seg_E"][j]; segLL[i] = g["seg_LL"][j]
        A = np.zeros((n, n)); E = np.zeros((2, n)); LL = 0.0
        for i in range(self.n_seg):  # hmm_add_expect order (khmm.c:346-359), LL += hmm_lk (em.c:48)
            LL += float(segLL[i])
            A += segA[i]
            E += segE[i][:2]
        return dict(A=A, E=E, LL=LL)
