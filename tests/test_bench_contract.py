"""bench.py's output contract: the one JSON line the driver reads, its roofline / cpu_baseline objects, the defaults
of the multi-GPU invocation.  CPU: argument defaults and the pure helpers; GPU: real (small) runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _parse(argv):
    bench = _bench()
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_defaults_follow_the_baseline_configs():
    a = _parse([])
    assert (a.gpus, a.scaling, a.log2n, a.kernel, a.dtype) == (1, "weak", 20, "matern32", "f64")      # c2
    a = _parse(["--gpus", "8"])
    assert (a.scaling, a.log2n, a.exchange) == ("strong", 24, "lib")                                    # c4, RCCL in libpgps
    a = _parse(["--gpus", "4", "--scaling", "weak"])
    assert (a.scaling, a.log2n) == ("weak", 20)
    a = _parse(["--gpus", "2", "--all-on-gpu0"])
    assert (a.exchange, a.dist_backend) == ("torch", "gloo")
    a = _parse(["--kernel", "rbf6", "--dtype", "f32"])
    assert a.kernel == "rbf6" and a.dtype == "f32"                                                      # c3


def test_vector_fp_lower_bound():
    bench = _bench()
    # c5: d = 11 fp64, 2^20 steps: 64 d^3 + 40 d^2 = 90 024 flop and 7048 B per step; flop-bound by that estimate
    v = bench.vector_fp(11, "f64", 1 << 20, 2.75, 7048)
    assert v["flops_per_step"] == 90024 and v["lower_bound_by"] == "vector_fp"
    assert abs(v["lower_bound_ms"] - 90024 * 2 ** 20 / 78.6e12 * 1e3) < 1e-12
    assert abs(v["frac"] - v["frac_of_lower_bound"]) < 1e-12 and 0.4 < v["frac"] < 0.5
    # c2: d = 2 fp64: byte-bound
    v = bench.vector_fp(2, "f64", 1 << 20, 0.0852, 280)
    assert v["lower_bound_by"] == "hbm" and abs(v["lower_bound_ms"] - 280 * 2 ** 20 / 8e12 * 1e3) < 1e-12


def test_kernels_by_name():
    bench = _bench()
    sys.path.insert(0, os.path.join(ROOT, "parallel-gps_amd"))
    assert bench.make_kernel("rbf6").get_sde().F.shape == (6, 6)
    assert bench.make_kernel("c5").get_sde().F.shape == (11, 11)
    assert bench.make_kernel("matern32").get_sde().F.shape == (2, 2)


def _run(args, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # ONE JSON line on stdout
    return json.loads(lines[0])


CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


@pytest.mark.gpu
def test_one_gpu_line_has_the_contract_fields():
    j = _run(["--log2n", "14", "--steps", "5", "--warmup", "2"])
    assert CONTRACT <= set(j) and "cpu_baseline" in j
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["scaling"] is None                     # one GPU: neither weak nor strong
    assert j["unit"] == "timesteps/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - (1 << 14) * 5 / (j["ms_per_step"] * 5e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_timed"] == 5                 # a short timed region samples every launch of the dominant kernel
    # every launch slot side by side (two slots within a few per cent swap the `dominant` role from box to box), and
    # whether the committed PMC traffic was measured on the kernels of this tree
    assert {"k_filter_reduce", "k_filter_apply", "k_smoother_apply"} <= set(r["slots"])
    for v in r["slots"].values():
        assert v["ms_per_pass"] > 0 and abs(v["frac"] - v["alg_bytes"] / (v["ms_per_pass"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
        assert v["frac"] <= 1.0
    assert "traffic_stale" in r
    assert j["f32_promoted"] is None and j["grid"] == "baseline"
    for leg in ("filter+smooth+log-lik", "log-lik only"):
        f = j["fused_path"][leg]
        assert f["ms_min"] <= f["ms_per_step"] <= f["ms_max"] and f["rounds"] >= 2
    assert set(j["fused_path"]["geometry"]) == {"lanes_per_workgroup", "steps_per_lane", "workgroups"}
    assert r["vector_fp"]["unit"] == "TFLOP/s"
    c = j["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["cores"] == 1
    assert c["value"] > 0 and c["unit"] == j["unit"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--kernel", "c5", "--log2n", "16"], ["--kernel", "rbf8", "--dtype", "f32", "--log2n", "16"],
                                  ["--kernel", "co2", "--log2n", "14"], ["--log2n", "19"]])
def test_no_launch_slot_is_priced_above_the_roofline(args):
    """Every slot of every kernel family is priced against the bytes of the contract its own kernels touch: the cooperative
    families' scan slots (chain totals only) carry no per-step bytes and report frac = None, never a fraction above 1
    (round 4's c5 line read 6.27 there); the resident launch (2^19 steps of d = 2) is priced against the whole contract."""
    j = _run(args + ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--main-only"])
    r = j["roofline"]
    assert r["slots"], r
    for name, v in r["slots"].items():
        assert v["frac"] is None or 0.0 < v["frac"] <= 1.0, (name, v)
        if v["frac"] is None:
            assert v["alg_bytes"] is None and name == "k_smoother_reduce"
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["whole_path_frac"] <= 1.0
    if args == ["--log2n", "19"]:
        assert r["kernel"] == "k_pkfs_resident" and r["kernel_family"] == 12 and set(r["slots"]) == {"k_pkfs_resident"}


@pytest.mark.gpu
def test_float32_line_says_whether_it_was_promoted():
    """BASELINE's c3 grid keeps float32 arithmetic; the reference's dense grid is promoted to fp64 arithmetic."""
    j = _run(["--kernel", "rbf6", "--dtype", "f32", "--log2n", "15", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--main-only"])
    assert j["dtype"] == "f32" and j["f32_promoted"] is False
    j = _run(["--kernel", "rbf6", "--dtype", "f32", "--log2n", "15", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--main-only",
              "--grid", "reference"])
    assert j["f32_promoted"] is True and j["grid"] == "reference"


@pytest.mark.gpu
def test_two_ranks_dry_run_reports_the_strong_scaling_line():
    """`bench.py --gpus 2` starts its own ranks; on a one-GPU box both use GPU 0 and gloo carries the two all-gathers."""
    j = _run(["--gpus", "2", "--all-on-gpu0", "--log2n", "15", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert CONTRACT <= set(j)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["steps_total"] == 1 << 15
    assert j["config"]["steps_per_gpu"] == 1 << 14
    assert abs(j["value"] - (1 << 15) * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    # the line checks itself: the same series on one GPU next to it, which exchange ran, how many ranks it saw
    ss = j["strong_scaling"]
    assert ss["one_gpu_ms"] > 0 and abs(ss["speedup"] - ss["one_gpu_ms"] / ss["n_gpu_ms"]) < 1e-9
    assert abs(ss["efficiency"] - ss["speedup"] / 2) < 1e-12
    assert j["rccl"]["in_library"] is False and j["exchange_fallback"] is False      # --all-on-gpu0 asks for torch / gloo


@pytest.mark.gpu
def test_segments_at_one_gpu_report_the_in_library_communicator():
    """`--force-segments` at one GPU runs the product's sharded path on a real RCCL communicator of size 1."""
    j = _run(["--force-segments", "--log2n", "15", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--main-only"])
    r = j["rccl"]
    assert r["in_library"] is True and r["ranks"] == 1 and j["exchange_fallback"] is False
    assert len(r["allgather_us"]) == 2 and all(0 < v < 1e4 for v in r["allgather_us"])
