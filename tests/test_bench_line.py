"""bench.py's contract line — the ONE stdout line the driver parses — stays small and parseable at every N.

r04's line had grown to 22 KB (per-kernel tables, per-rank arrays, prose) and the driver recorded `parsed: null`.
The line is now built from a whitelist of scalars (bench.contract_line, < 4 KB); everything else goes to
gpurun_out/bench_extras_n{N}.json.  CPU half: the builder on recorded full outputs of earlier rounds and on a
synthetic worst case.  GPU half: the real command the driver runs, and the 2-rank gloo path."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config"}


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _scalars_only(obj, depth=0):
    """No lists anywhere; dict nesting at most 3 deep; strings short."""
    assert depth <= 3
    if isinstance(obj, dict):
        for v in obj.values():
            _scalars_only(v, depth + 1)
    else:
        assert not isinstance(obj, (list, tuple)), obj
        assert not isinstance(obj, str) or len(obj) <= 160, obj


@pytest.mark.parametrize("recorded", ["r04l_bench_line.json", "r04_bench_n2_gloo_line.json", "r03_bench_n2_gloo_line.json"])
def test_contract_line_from_recorded_full_outputs(recorded):
    """The full dicts earlier rounds printed (10-22 KB, committed under profiles/) through today's builder."""
    path = os.path.join(ROOT, "profiles", recorded)
    if not os.path.exists(path):
        pytest.skip(f"{recorded} not committed")
    raw = open(path).read().strip()
    full = json.loads(raw if raw.startswith("{\n") else raw.splitlines()[-1])
    text = _bench().contract_line(full, "gpurun_out/bench_extras_n1.json")
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    assert CONTRACT_KEYS <= set(line) and line["value"] == pytest.approx(full["value"], rel=1e-6)
    assert line["n_gpus"] == full["n_gpus"] and line["steps"] == full["steps"] and line["warmup"] == full["warmup"]
    _scalars_only(line)
    if full["n_gpus"] == 1 and "roofline" in full:
        rf = line["roofline"]
        # the line alone lets a reader recompute both fractions
        assert rf["frac"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 8e12,
                                           rel=1e-4)
        if rf["avg_launch_ms_cold"]:
            assert rf["frac_hbm_cold"] == pytest.approx(
                rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms_cold"] * 1e-3) / 8e12, rel=1e-4)
        assert rf["traffic_counts"] == "l2_fabric_bytes" and rf["bound"] == "hbm"
    if "cpu_baseline" in full:
        assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["cores"] >= 1


def test_contract_line_worst_case_stays_under_4k():
    """Eight ranks, every optional object present and bloated: the line still fits and still parses."""
    big = "x" * 5000
    out = {
        "metric": big, "value": 123456.789012345, "unit": "RK-stages/s", "n_gpus": 8, "steps": 20, "warmup": 5,
        "ms_per_step": 0.0812345678, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": big, "global_batch": 65536, "rows_per_gpu": 8192, "dim": 128, "parallelism": big,
                   "accepted": 10, "rejected": 0, "lookahead": True, "hip_graph": True, "backend": "nccl",
                   "nested": {"dropped": big}, "listed": [big] * 8},
        "blocks": {"ms_per_step": {"min": 0.08, "median": 0.081, "max": 0.09, "blocks": [0.08] * 5},
                   "per_rank_ms_per_step": [0.08 + 0.001 * i for i in range(8)], "value_is": big},
        "roofline": {"bound": "hbm", "kernel": big, "achieved": 5500.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6875,
                     "algorithmic_bytes_per_launch": 29360128, "avg_launch_ms": 0.00534, "traffic": None, "note": big,
                     "timing": big},
        "breakdown": {"per_rank": [{"rank": i, "top_kernels": {big + str(j): {"avg_us": 1.0} for j in range(8)}}
                                   for i in range(8)]},
        "weak": {"value": 1.0, "ms_per_step": 1.0, "config": {"workload": big}, "blocks": {}},
        "strong": {"value": 1.0, "ms_per_step": 1.0},
        "lockstep": {"value": 1.0, "ms_per_step": 1.0, "collective": big},
        "adjoint": {m: {"ms_per_pass": 1.0, "rk_stages_per_s": 2.0, "nfe_fwd": 20, "nfe_bwd": 86,
                        "allreduce": {"calls": 1, "bytes": 395520, "ms": 0.1, "what": big}, "breakdown": {"x": big}}
                    for m in ("strong", "weak", "strong_hip_graph_auto")},
        "comm": {"devices": [{"rank": i, "device_name": big} for i in range(8)]},
        "rccl_ranks": 8, "backend": "nccl", "rel_err": 2.4e-6, "rel_err_vs_reference": None, "nfe": 68,
        "reference_nfe": None, "note": big, "extras_s": {"a": 1.0},
    }
    text = _bench().contract_line(out, "gpurun_out/bench_extras_n8.json")
    assert len(text) < 4096
    line = json.loads(text)
    _scalars_only(line)
    assert line["rank_ms_per_step"] == {"min": 0.08, "max": 0.087}
    assert "breakdown" not in line and "comm" not in line and "nested" not in line["config"]


def test_error_line_is_small_and_parseable(capsys):
    import argparse
    b = _bench()
    b.error_line(argparse.Namespace(gpus=8, steps=20, warmup=5), "nope", visible_devices=1)
    text = capsys.readouterr().out.strip()
    assert len(text) < 4096 and json.loads(text)["value"] is None


@pytest.mark.gpu
def test_driver_command_prints_one_small_line(tmp_path):
    """`python bench.py --gpus 1 --steps 2 --warmup 1` (the driver's command shape): exactly one stdout line that is
    JSON, it is the LAST line, < 4 KB, with `roofline` and `cpu_baseline`; the extras file exists and holds the rest."""
    # (the TunableOp child of the extras is a second bench process of ~1 min: not in the test suite)
    env = dict(os.environ, TDEQ_BENCH_EXTRAS_DIR=str(tmp_path), TDEQ_BENCH_TUNABLEOP="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    stdout_lines = r.stdout.strip().splitlines()
    json_lines = [ln for ln in stdout_lines if ln.startswith("{")]
    assert len(json_lines) == 1 and stdout_lines[-1] == json_lines[0]
    assert len(json_lines[0]) < 4096
    line = json.loads(json_lines[0])
    assert json.loads(json.dumps(line)) == line
    assert CONTRACT_KEYS <= set(line) and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and line["scaling"] == "weak" and line["dtype"] == "f32"
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and 0 < rf["frac_hbm_cold"] < 1 and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 8e12, rel=1e-4)
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"
    assert line["nfe"] == line["reference_nfe"] == 68 and line["rel_err_vs_reference"] < 1e-5
    _scalars_only(line)
    extras = json.load(open(os.path.join(tmp_path, "bench_extras_n1.json")))
    assert {"configs", "shard_regime", "adjoint_full", "solver_only", "extras_s", "low_precision", "vector_tolerances"} <= set(extras)
    assert extras["low_precision"]["bf16"]["hip_kernels"]["backend"] == "hip-low"
    assert extras["vector_tolerances"]["vector_rtol_fused"]["fused_norm"] is True
