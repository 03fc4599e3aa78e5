"""The driver's contract with bench.py, checked on the GPU box: `python bench.py --steps K --warmup W` prints ONE JSON line
on stdout with the agreed keys (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload + the `roofline` and `cpu_baseline` objects), `value` consistent with
`ms_per_step`, and the roofline fraction a fraction."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    # (--no-batch1: without the information-only legs `batch1` / `c3_regime` / `c4` / `config5`, which are best-effort
    # objects behind try / except -- tools/profile_r05.sh runs the full line; this test keeps the GPU suite short)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-iters", "1", "--cpu-pairs", "1",
                        "--no-batch1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "BASELINE.json configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    pairs = d["config"]["pairs_per_gpu_per_step"]
    assert abs(d["value"] - pairs / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]  # value = pairs / time
    assert 300 < d["value"] < 5000
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3
    assert 0.3 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["launches_timed"] == 18 * 3  # live HIP events around every attention launch of the timed region
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert d["self_check"]["pairs_equal"] is True and d["self_check"]["keypoint_sets_equal"] is True
    assert not any(k in d for k in ("batch1", "c3_regime", "c4", "config5"))  # skipped by --no-batch1
    # every rank's own numbers ride on the line (one row at N = 1): the N > 1 line explains itself
    per = d["config"]["per_rank"]
    assert len(per) == 1 and per[0]["rank"] == 0 and d["config"]["rccl_ranks"] == 1
    assert 0 < per[0]["busy_s"] <= per[0]["elapsed_s"] and abs(per[0]["elapsed_s"] - d["ms_per_step"] * 3e-3) < 1e-3
    assert abs(per[0]["attention_avg_launch_ms"] - roof["avg_launch_ms"]) < 1e-3 and per[0]["stem_avg_launch_ms"] > 0
    assert per[0]["probe_tflops"] == roof["sustained_mfma_probe"]["tflops"] and 1.0 < per[0]["probe_shader_clock_ghz"] < 2.6
    assert d["config"]["per_rank_summary"]["busy_s"] == [per[0]["busy_s"]] * 3
