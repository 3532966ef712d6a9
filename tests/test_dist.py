"""Multi-process sharding of the E-step (psmc_amd/dist.py) on CPU: world_size 2,
gloo.  The per-rank engine is the oracle (test infrastructure) so that what is
exercised is the product's partitioning, gather order and reduction -- the same
code that runs one process per GPU over RCCL on the node."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import orc
    from psmc_amd.dist import ShardedEStep
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
    a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
    s = np.load(os.path.join(ROOT, "tests", "golden", "segments_small.npz"))
    segs = [s[k] for k in sorted(s)]
    O = orc.Oracle()

    class OracleEngine:
        def __init__(self, n, local, mode_):
            self.local = local
        def estep(self, a, e, a0):
            return O.estep(a, e, a0, self.local)
        def estep_segments(self, a, e, a0):
            return O.estep(a, e, a0, self.local, per_seg=True)

    sh = ShardedEStep(64, segs, mode=mode, make_engine=lambda n, l, m: OracleEngine(n, l, m))
    r = sh.estep(a, e, a0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=r["A"], E=r["E"], LL=r["LL"], mine=np.array(sh.mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_two_ranks_match_single_process(tmp_path, golden, oracle, mode):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "exact" else 1)
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    p = golden.params("n64_curve")
    ref = oracle.estep(p["a"], p["e"], p["a0"], golden.segs_small)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == list(range(len(golden.segs_small)))
    for r in (r0, r1):
        if mode == "exact":  # ordered sum on every rank: bit-identical to the serial loop of em.c:36-55
            assert np.array_equal(r["A"], ref["A"]) and np.array_equal(r["E"], ref["E"]) and float(r["LL"]) == ref["LL"]
        else:
            assert np.abs(r["A"] - ref["A"]).max() <= 1e-12 * np.abs(ref["A"]).max()
            assert abs(float(r["LL"]) - ref["LL"]) <= 1e-12 * abs(ref["LL"])
    assert np.array_equal(r0["A"], r1["A"])


def test_partition_is_balanced_and_complete():
    from psmc_amd.dist import partition_segments
    from psmc_amd import sim
    lens = sim.human_like_lengths(30_000_000, 90)
    for world in (1, 2, 4, 8):
        parts = partition_segments(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(90))
        load = np.array([lens[p].sum() for p in parts])
        assert load.max() <= 1.08 * load.mean() or world == 1


def _hip_worker(rank, world, port, mode, out_dir):
    """One process per rank, BOTH on GPU 0 (the test box has one), gloo rendezvous: the product's ShardedEStep with
    its default engine -- the HIP library -- so partition, per-rank E-step on the device and the collective all run."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["LOCAL_RANK"] = "0"
    import torch.distributed as dist
    from psmc_amd.dist import ShardedEStep
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
    a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
    s = np.load(os.path.join(ROOT, "tests", "golden", "segments_mid.npz"))
    segs = [s[k] for k in sorted(s)]
    sh = ShardedEStep(64, segs, mode=mode)
    assert sh.device == 0 and sh.engine is not None
    r = sh.estep(a, e, a0)
    r2 = sh.estep(a, e, a0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=r["A"], E=r["E"], LL=r["LL"], A2=r2["A"], mine=np.array(sh.mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_two_ranks_on_one_gpu_with_the_hip_engine(tmp_path, golden, mode):
    """world_size 2 with the HIP engine on both ranks (not the oracle): exact = the golden of the reference bit for bit
    on every rank; fast = within tolerance, identical on both ranks (they reduce the same two vectors)."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 2000) + (0 if mode == "exact" else 1)
    mp.spawn(_hip_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    gm = golden.mid
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == list(range(6)) and len(r0["mine"]) > 0 and len(r1["mine"]) > 0
    for r in (r0, r1):
        if mode == "exact":
            assert np.array_equal(r["A"], gm["n64_curve.A"]) and np.array_equal(r["E"], gm["n64_curve.E"]) and float(r["LL"]) == float(gm["n64_curve.LL"])
        else:
            assert np.abs(r["A"] - gm["n64_curve.A"]).max() <= 1e-10 * np.abs(gm["n64_curve.A"]).max()
            assert abs(float(r["LL"]) - float(gm["n64_curve.LL"])) <= 1e-12 * abs(float(gm["n64_curve.LL"]))
    assert np.array_equal(r0["A"], r1["A"]) and np.array_equal(r0["A2"], r1["A2"])
