"""The driver's contract for bench.py's JSON line, checked on the committed line of the final tree
(profiles/r05_bench_line.json): the keys the driver parses, the `roofline` / `cpu_baseline` objects of this tier, the
consistency of the numbers with each other (value = steps / time, frac = achieved / peak, the algorithmic flop of the dominant
launch), and that bench.py itself parses and keeps its flags.  No GPU needed."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_honours_the_contract():
    r = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["higher_is_better"] is True and r["dtype"] == "f16" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"] and "configs[1]" in r["config"]["workload"]
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) < 0.01 * r["value"]
    ro = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and ro["peak"] == 2500.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    # algorithmic flop of the dominant launch: 4 B HW M C with B = 16, HW = 4096, C = 320 and M from the line itself
    M = r["config"]["cross_frame_keys_M"]["L3"]
    assert abs(ro["algorithmic_flop_per_launch"] - 4.0 * 16 * 4096 * M * 320) < 1.0
    assert abs(ro["achieved"] - ro["algorithmic_flop_per_launch"] / (ro["avg_launch_us"] * 1e-6) / 1e12) < 0.5
    assert ro["traffic"] > ro["algorithmic_bytes_per_launch"] and ro["traffic_source"].startswith("r05_")  # this round's PMC file
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["unit"] == r["unit"]
    # the auxiliary legs of this round
    assert r["cfg3"]["ms_per_step"] > 0 and "per_layer_stats" in r["cfg3"] and "dropped" not in r["cfg3"]["timing"].replace("nothing dropped", "")
    ld = r["latent_delta"]
    for tag in ("fp16_latents", "fp32_latents"):
        for d, n in zip(ld[tag]["max_abs_delta_per_step"], ld[tag]["reference_own_fp16_noise_per_step"]):
            assert d <= 1.5 * n + 1e-3
    assert r["full_step"]["fresco_amd_schedule_mean_ms"] < r["full_step"]["reference_torch_path_schedule_mean_ms"]
    assert r["f3_gmflow"]["dense_layers"]["launches"] > 0


def test_bench_py_parses_and_keeps_its_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert "dist.barrier()" in src and "torch.cuda.synchronize()" in src and "ReduceOp.MAX" in src
