"""The one-line JSON contract of bench.py, checked on the line the last GPU run committed under profiles/
(bench.py itself needs a GPU; the schema does not)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1])), files[-1]


def test_bench_line_has_the_contract_fields():
    d, path = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (k, path)
    assert d["unit"] == "graphs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / mean launch time
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["mean_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    # 4nF + 4nF + 4E + 4(n+1) per graph-layer
    n, E, B = d["config"]["nodes_per_graph"], d["config"]["edges_per_graph"], d["config"]["graphs_per_gpu"]
    assert r["algorithmic_bytes_per_launch"] == (8 * n * 64 + 4 * E + 4 * (n + 1)) * B
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
