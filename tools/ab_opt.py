"""Same-box A/B of the launch-form switches of opt_fast.hip (read per call by the library): one process, one set of
inputs, every configuration measured with tools/bench_opt.measure (1 warm-up + 3 timed runs per layer; per-kernel HIP
events from an instrumented single-stream run).  usage: python tools/ab_opt.py [rounds]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_opt

SWITCHES = tuple(os.environ.get("AB_SWITCHES", "FRESCO_GRAM_Z").split(","))  # the library reads them per call
SWITCHES = tuple(k for k in SWITCHES if k != "NONE")
CONFIGS = [("default", {})] + [(k + "=0", {k: "0"}) for k in SWITCHES] + ([("all=0", {k: "0" for k in SWITCHES})] if len(SWITCHES) > 1 else [])

if os.environ.get("AB_CONFIGS"):  # "name:K=V,K=V;name2:K=V": explicit configurations beside the default
    CONFIGS = [("default", {})]
    for item in os.environ["AB_CONFIGS"].split(";"):
        name, kv = item.split(":")
        CONFIGS.append((name, dict(x.split("=") for x in kv.split(","))))
    SWITCHES = tuple(sorted({k for _, e in CONFIGS for k in e}))

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for rnd in range(rounds):
        for name, env in CONFIGS:
            for k in SWITCHES:
                os.environ.pop(k, None)
            os.environ.update(env)
            r = bench_opt.measure(int(os.environ.get("AB_ITERS", "20")), verbose=False, baselines=False)
            kern = {L: {k: v for k, v in d.items() if not k.endswith("_roofline")} for L, d in r["kernel_avg_us"].items()}
            print("round %d %-22s cfg3 %.2f ms  layers %s  kernels %s" % (rnd, name, r["ms_per_step"], r["per_layer_ms"],
                                                                        json.dumps(kern)), flush=True)
