"""The committed bench line (profiles/r02_bench.json, an un-edited output of `python bench.py` on an MI355X box) carries every
field the driver's contract names, with consistent values.  CPU only: it guards the shape of the line, not the numbers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "sweeps/s" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["vs_baseline"] is None                                   # BASELINE.md holds no published number for this metric
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6   # one step = one full solve of one sweep
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["traffic"] is None or r["traffic"] < r["algorithmic_bytes_per_launch"]      # the working set is cache resident
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert d["value"] > 100 * c["value"]
    names = [x["name"] for x in d["configs"]]
    assert names[:4] == ["C1", "C2", "C3", "C4"] and all(x["parity"]["ok"] for x in d["configs"])
    assert d["parity"]["iterations_gpu"] == d["parity"]["iterations_oracle"]
    assert isinstance(b.get("metric", ""), str)
