"""bench.py's JSON line (the driver's contract) assembled from fake measurements for every plan the library can
report -- no GPU: keys, units, the roofline object for whichever kernel is the longest, the fields the judge reads."""
import argparse
import json
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERN = dict(total=14.4, chains=14.3, tail=0.07, expect=6.9, reduce=0.07, fwd_sweep=7.2, bwd_sweep=14.1, forward=14.3, backward=0.07)
DIAG = dict(warm_err_fwd=5e-13, warm_err_bwd=4e-13, n_chunks=8127, warmup=3072, fwd_rounds=0, bwd_rounds=0, fwd_tiles=0, bwd_tiles=0,
            structured=True, tile_len=3712, items_fwd=7948, items_bwd=7979, back_half=1, ckpt=False, fused_launches=2, merged_phase1=0)


@pytest.mark.parametrize("case", ["fused_fwd_longest", "fused_counts_longest", "unfused_struct", "dense", "exact"])
def test_bench_line_contract(case):
    import bench
    args = argparse.Namespace(steps=20, warmup=5, scaling="weak", mode="exact" if case == "exact" else "fast", fixed_params=0)
    kern, diag, fast = dict(KERN), dict(DIAG), case != "exact"
    if case == "fused_counts_longest":
        kern.update(fwd_sweep=5.8, expect=6.8)
    elif case == "unfused_struct":
        diag.update(back_half=0)
    elif case == "dense":
        diag.update(back_half=0, structured=False)
    elif case == "exact":
        kern, diag = dict(total=4028.0, forward=2190.0, backward=1425.0, expect=412.0, reduce=0.004, fwd_sweep=0.0, bwd_sweep=0.0), {}
    lens = np.array([2490000, 2429561] + [285000] * 88)
    bins = int(lens.sum())
    out = bench.make_line(args, fast, 1, bins, bins, lens, 90, kern, diag, bins / 14.65e-3, 14.65, "test trajectory", 25, 4225)
    json.dumps(out)  # serialisable
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["unit"] == "bins/s" and out["dtype"] == "f64" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["longest_segment"] == 2490000
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    if case == "fused_fwd_longest":
        assert r["kernel"] == "k_fwd_struct<speculate>" and r["bound"] == "hbm" and r["also"]["kernel"] == "k_bwd_count4f_struct"
        assert 0.3 < r["also"]["frac"] < 0.7 and 0.3 < r["also"]["mfma_cycles_frac"] < 0.7
        assert abs(r["achieved"] - bins * 521 / 7.2e-3 / 1e9) < 1e-6
    if case == "fused_counts_longest":
        assert r["kernel"] == "k_bwd_count4f_struct" and r["bound"] == "mfma" and r["also"]["kernel"].startswith("k_fwd_struct")
        assert abs(r["achieved"] - bins * (2 * 64 * 64 + 24 * 64) / 6.8e-3 / 1e12) < 1e-9


def test_replayed_traffic_is_guarded_by_the_kernel_sources(tmp_path):
    """roofline.traffic is replayed from a committed rocprofv3 --pmc record: only a record taken from the kernel sources of
    this build may be quoted -- a record without a hash, or with another one, gives null and says why."""
    import bench
    rec = {"bins": 30000001, "kernels": {"k_bwd_count4f_struct": {"hbm_bytes_per_launch": 7.9e9}}, "kernel_sources": list(bench.TRAFFIC_SOURCES)}
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(rec))
    t, note = bench.replayed_traffic(str(p), 30000001, "k_bwd_count4f_struct")
    assert t is None and "no kernel_src_sha16" in note
    rec["kernel_src_sha16"] = "0123456789abcdef"
    p.write_text(json.dumps(rec))
    t, note = bench.replayed_traffic(str(p), 30000001, "k_bwd_count4f_struct")
    assert t is None and "other kernel sources" in note
    rec["kernel_src_sha16"] = bench.kernel_src_sha16(rec["kernel_sources"])
    p.write_text(json.dumps(rec))
    t, note = bench.replayed_traffic(str(p), 30000001, "k_bwd_count4f_struct")
    assert t == 7.9e9 and "same kernel sources" in note
    assert bench.replayed_traffic(str(p), 123, "k_bwd_count4f_struct")[0] is None          # another workload
    assert bench.replayed_traffic(str(p), 30000001, "k_nothing")[0] is None                 # another kernel
    assert bench.replayed_traffic(str(tmp_path / "absent.json"), 30000001, "k")[0] is None
