"""The bench line recorded on the MI355X (profiles/r03_bench_line.txt) carries every field of the driver's contract."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_fields():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_line.txt")).read())
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "samples/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]            # whole-job pairs/s
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    assert 0 < r["mfma"]["frac"] <= 1.0 and 0 < r["hbm"]["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "samples/s" and c["sample"]
    assert d["s1_classifier_step"]["value"] > 0 and set(d["north_star_extras"]) == {"omni_scale_fe_forward", "cpc_cross_gram", "cdan_random_layer_gemm"}
    assert d["f32_mode"]["value"] > 0 and d["f32_mode"]["dtype"] == "f32" and d["dtype"] == "bf16x3"      # the exact-f32 figure sits beside it
    anomaly = c["anomaly_mode_on"] if "anomaly_mode_on" in c else c["small_sample"]["anomaly_mode_on"]
    assert anomaly["value"] > 0 and d["mode"] == "graph" and d["config"]["sources"] == 1 and d["config"]["c_in"] == 1
