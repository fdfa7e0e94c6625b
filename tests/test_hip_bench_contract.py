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
    # BASELINE configs[2..4]: short fenced passes behind (and outside) the headline's timed region, in the same line
    sec = j["secondary_configs"]
    assert len(sec) == 3 and [e["workload"].split(":")[0] for e in sec] == [f"BASELINE configs[{i}]" for i in (2, 3, 4)]
    for e in sec:      # structure; a child that timed out on a cold / busy box reports `error` or `skipped` instead of numbers
        assert "workload" in e and "flags" in e
        if "error" in e or "skipped" in e:
            continue
        for k in ("value", "ms_per_step", "step_mfma_fraction", "steps"):
            assert k in e, k
        assert e["value"] > 0 and abs(e["value"] - e["per_gpu_batch"] * 1e3 / e["ms_per_step"]) / e["value"] < 1e-3
        assert 0.02 < e["step_mfma_fraction"] < 1.0
    assert sum(1 for e in sec if "value" in e) >= 1
    assert j["secondary_configs_wall_s"] < 340          # bench.SECONDARY_BUDGET_S + one child's start-up
    # the record is tied to the binary, and carries the dominant kernel's numbers per problem
    assert "src:" in j["library"] and "git:" in j["library"]
    rf = j["roofline"]
    assert rf["traffic"] is None or rf["traffic"]["lib_src_hash"] in j["library"]
    assert rf["traffic"] is not None or "traffic_note" in rf
    names = {k["name"] for k in rf["per_kernel"]}
    assert {"image QKV", "image out-projection", "image MLP up + GELU", "image MLP down"} <= names, names
    for k in rf["per_kernel"]:
        for key in ("name", "M", "N", "K", "launches_per_step", "avg_us", "frac", "tflops"):
            assert key in k
        assert 0.0 < k["frac"] < 1.0 and k["avg_us"] > 0
        if k["name"].startswith("image "):
            assert k["M"] == 256 * 197 and abs(k["launches_per_step"] - (11 if k["name"] != "image QKV" else 12)) <= 1


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """N > 1 readiness without an N-GPU node: `python bench.py --gpus 2` spawns its own two ranks; with
    MVLPT_DEBUG_SHARE_GPU=1 both sit on cuda:0 and the collectives go through gloo.  Everything else is the real path: the
    self-spawn, rendezvous, parameter broadcast, barrier + synchronize timing protocol with max over ranks, the flat gradient
    all-reduce, and ONE JSON line from rank 0 with the whole-job rate."""
    env = dict(os.environ, MVLPT_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--batch", "64",
                          "--no-cpu-baseline", "--no-trim-extra"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 3 and j["scaling"] == "weak"
    assert j["config"]["parallelism"] == "dp2" and j["config"]["per_gpu_batch"] == 64 and j["config"]["global_batch"] == 128
    assert j["config"]["text_tower"] == "replicated per GPU"                       # 100 classes: below the sharding threshold
    assert abs(j["value"] - 128 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3      # whole-job rate over BOTH ranks
    assert j["config"]["loss"] == j["config"]["loss"] and "cpu_baseline" not in j
    # who took part: one record per rank; in this debug mode both sit on the one device and say so
    ranks = j["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and len({r["pid"] for r in ranks}) == 2
    assert all(k in ranks[0] for k in ("local_device", "device_name", "device_uuid", "pci_bus_id", "backend"))
    assert j["config"]["distinct_devices"] == 1 and "debug_shared_gpu" in j["config"]


@pytest.mark.gpu
def test_bench_two_ranks_shard_the_text_tower_by_default_for_many_classes():
    """1000 classes on 2 ranks: the text tower is class-sharded without being asked (bench.py --shard-text default)."""
    env = dict(os.environ, MVLPT_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "32",
                          "--classes", "1000", "--no-cpu-baseline", "--no-trim-extra", "--no-kernel-timing"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["text_tower"] == "class-sharded over ranks"
