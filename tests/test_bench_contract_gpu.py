"""bench.py's output contract on the GPU box: stdout is exactly ONE JSON line carrying the driver's keys, `roofline` and (at N=1)
`cpu_baseline`; checked on a 1-step run of the default workload (c3), of configs[4] at its full length (c5: one 256-frame
video, 31 windows) and of the clip-preparation workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must be one JSON line, got {len(lines)}: {out.stdout[:500]}"
    return json.loads(lines[0])


def _check_common(r, steps, warmup):
    assert KEYS <= set(r), sorted(KEYS - set(r))
    assert r["n_gpus"] == 1 and r["steps"] == steps and r["warmup"] == warmup and r["higher_is_better"] is True
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["vs_baseline"] is None and "workload" in r["config"]
    roof = r["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3


def test_default_workload_line(dev):
    r = _run(["--steps", "1", "--warmup", "1", "--cpu-baseline-quick"])  # (the CPU leg on a scaled sample: the test is about the line)
    _check_common(r, 1, 1)
    assert r["unit"] == "frames/s" and r["dtype"] == "bf16" and r["scaling"] == "weak" and "all heads" in r["metric"]
    assert r["roofline"]["bound"] == "mfma" and r["roofline"]["peak"] == 2500.0 and "roofline_attention" in r
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "frames/s" and cb["sample"]
    # frames per step of configs[2]: 4 clips x 16 frames
    assert abs(r["value"] - 64 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-2


def test_prep_workload_line(dev):
    r = _run(["--workload", "prep", "--steps", "3", "--warmup", "1"])
    _check_common(r, 3, 1)
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["peak"] == 8000.0 and r["cpu_baseline"]["kind"] == "port"


def test_c5_workload_line_at_full_length(dev):
    """configs[4] on one GPU at its own length: 256 frames = 31 windows, all heads, on-GPU alignment.  Property checks at full
    size: the line carries the pieces of the step (encoders, decoders, replicated dense stitch, tracker recursion in full and on
    an eighth of the queries) and what they imply for 8 GPUs with the tracker after / beside the decoders."""
    r = _run(["--workload", "c5", "--steps", "1", "--warmup", "1", "--cpu-baseline-quick"])
    _check_common(r, 1, 1)
    assert r["scaling"] == "strong" and "31 overlapping" in r["config"]["workload"]
    assert abs(r["value"] - 256 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-2
    assert 0 < r["phase3_ms"] < r["ms_per_step"] and 0 < r["phase1_ms"] < r["ms_per_step"]
    assert abs(r["phase1_ms"] - (r["phase1a_encoder_ms"] + r["phase1b_decoders_ms"])) < 0.01
    # the tracker runs beside the decoders in the timed step: the step is shorter than the pieces one after the other
    assert r["ms_per_step"] < r["serial_sum_ms"] * 1.02
    t8 = r["phase1a_encoder_ms"] / 8 + max(r["phase3_track_ms_on_an_eighth_of_the_queries"], r["phase1b_decoders_ms"] / 8) + r["phase3_dense_ms"]
    assert abs(r["implied_8gpu_ms_tracker_beside_decoders"] - t8) < 0.01
    assert abs(r["implied_8gpu_speedup_tracker_beside_decoders"] - r["ms_per_step"] / t8) < 0.01
    assert r["implied_8gpu_speedup_tracker_beside_decoders"] >= r["implied_8gpu_speedup_tracker_after_decoders"]


def test_host_io_pass_is_reported_beside_value_never_as_value(dev):
    """--host-io (c2, three steps): the PCIe-inclusive pass - inputs from pinned host memory, every output copied back inside the
    step - is its own object of the line; `value` stays the resident-input rate and is not slower than it."""
    r = _run(["--workload", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--host-io"])
    _check_common(r, 3, 1)
    pc = r["pcie_inclusive"]
    assert pc["h2d_bytes_per_step"] >= 3 * 16 * 224 * 224 * 4 and pc["d2h_bytes_per_step"] >= 16 * 224 * 224 * 4
    assert pc["value"] > 0 and pc["ms_per_step"] > 0
    assert abs(r["value"] - 16 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-2
    assert pc["value"] <= r["value"] * 1.05
