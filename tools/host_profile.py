"""Where the HOST time of a config-2 step goes (cProfile over eager steps; the GPU runs behind): python tools/host_profile.py [steps]"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
layers, params = B.build_workload(8, 512, dev)
proc, ctrl, refs, paras, masks = B.make_processor(layers, params, dev, None)
for l in layers:
    l["hidden_local"], l["ref_local"] = l["hidden"], l["ref"]
refs = [l["ref_local"] for l in layers]
with torch.no_grad():
    for mode in sorted(set(B.SCHEDULE)):
        B.run_step(proc, ctrl, layers, mode, refs, paras, masks)
    for s in range(6):
        B.run_step(proc, ctrl, layers, B.SCHEDULE[s % len(B.SCHEDULE)], refs, paras, masks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        B.run_step(proc, ctrl, layers, B.SCHEDULE[s % len(B.SCHEDULE)], refs, paras, masks)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("issue %.3f ms/step, wall %.3f ms/step" % (1e3 * t_issue / steps, 1e3 * t_all / steps))
    pr = cProfile.Profile()
    pr.enable()
    for s in range(steps):
        B.run_step(proc, ctrl, layers, B.SCHEDULE[s % len(B.SCHEDULE)], refs, paras, masks)
    pr.disable()
    torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
