"""bench.py prints ONE JSON line with the driver's contract fields (GPU box: the HIP path is the thing measured).

Small K / W so that the test stays under a minute; the CPU baseline leg is exercised once on a tiny sample by the default
`python bench.py` run, not here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "4", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in j, k
    assert j["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 4 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["dtype"] == "fp16" and j["unit"] == "images/sec"
    assert "workload" in j["config"] and "BASELINE configs[1]" in j["config"]["workload"] and "model" not in j["config"]
    assert j["config"]["text_positions_evaluated"] == 77                    # the headline evaluates every text position
    assert abs(j["value"] - 256 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3   # whole-job images/s = batch / step time
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.05 < r["frac"] < 1.0 and r["launches_per_step"] > 100
    # extras: the exact trim-to-EOT rate is reported beside, never as, the headline
    assert j["text_trimmed_to_eot"]["text_positions_evaluated"] < 77 and j["text_trimmed_to_eot"]["value"] > j["value"] * 0.9
    if "clock" in j:
        assert 100.0 < j["clock"]["sclk_mhz_avg"] <= 2500.0 and j["clock"]["samples"] >= 2
