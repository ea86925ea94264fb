"""time the fused K | V projection + pack launch (and the two-launch path) at config 2's shapes: python tools/bench_kvproj.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import fresco_amd
from fresco_amd import _lib

def main():
    dev = torch.device("cuda")
    lib = _lib.load()
    layers, params = bench.build_workload(8, 512, dev)
    for l in layers:
        l["hidden_local"], l["ref_local"] = l["hidden"], l["ref"]
    proc, ctrl, refs, paras, masks = bench.make_processor(layers, params, dev)
    out = {}
    for fused in (True, False):
        proc.fuse_kv_pack = fused
        with torch.no_grad():
            for _ in range(3):
                bench.run_step(proc, ctrl, layers, "cf", refs, paras, masks)
            torch.cuda.synchronize()
            lib.fresco_prof_enable(4096)
            for _ in range(10):
                bench.run_step(proc, ctrl, layers, "cf", refs, paras, masks)
            torch.cuda.synchronize()
            lib.fresco_prof_disable()
        agg = {}
        for tag, d, ms in bench.read_prof(lib, 4096):
            if tag in (2, 10):
                agg.setdefault((tag, d), []).append(ms)
        out[fused] = {("%s%s" % ({2: "kv_pack", 10: "linear"}[k[0]], list(k[1]))): round(1e3 * sum(v) / len(v), 2) for k, v in sorted(agg.items())}
    print("lib", os.environ.get("FRESCO_HIP_LIB", "default"))
    for f, d in out.items():
        print(" fused" if f else " two-launch", d)

if __name__ == "__main__":
    main()
