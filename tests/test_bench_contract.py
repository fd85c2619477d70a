"""The one JSON line bench.py prints: every field of the driver's contract is there and consistent."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_leg_reports_its_sample():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline(3, 30, sample_b=4, reps=1)
    assert out["kind"] == "port" and out["unit"] == "molecules/s" and out["value"] > 0
    assert out["cores"] >= 1 and "4 molecules" in out["sample"] and out["cpu"]


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "128",
           "--cpu-sample", "4"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == "encoder fwd+bwd molecules/sec" and d["unit"] == "molecules/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    dom = max(d["kernels"], key=lambda k: k["ms_per_step"])
    assert dom["kernel"] == r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "molecules/s" and c["sample"]
