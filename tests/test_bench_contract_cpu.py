"""The bench line's contract (the driver parses it): the committed line of the final tree carries every required key, the roofline
and cpu_baseline objects, and numbers that are consistent with each other."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    # rNN_<tag>_bench.json is bench.py's line (rNN_<tag>_train_bench.json etc. belong to the other benchmarks); the newest by name
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")) if re.fullmatch(r"r\d+_[a-z0-9]+_bench\.json", os.path.basename(f)))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1])), files[-1]


def test_committed_bench_line_has_the_contract_keys():
    line, path = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, (k, path)
    assert line["unit"] == "cells/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert line["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == "cells/s"


def test_committed_bench_line_is_self_consistent():
    line, _ = _latest()
    cells = line["config"]["global_cells"]
    assert abs(line["value"] - cells / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    # the kernels' device time does not exceed the step, and the dominant kernel is the one the roofline prices
    per_step = sum(v["ms"] for v in line["kernels"].values()) / line["steps"]
    assert per_step <= line["ms_per_step"] * 1.001
    dom = max(line["kernels"].items(), key=lambda kv: kv[1]["ms"])[0]
    assert line["roofline"]["kernel"] == dom
    # measured HBM bytes per launch of the dominant kernel are not below its algorithmic bytes (and within 1 % of them)
    t, a = line["roofline"]["traffic"], line["roofline"]["algorithmic_bytes_per_launch"]
    assert t is None or a <= t <= 1.01 * a
