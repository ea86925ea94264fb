"""fill the @PLACEHOLDER@ numbers of DESIGN.md's status table from a bench line: python tools/fill_design.py profiles/r06_bench_line.json [n_gpu_tests] [rocprof_flash_us]"""
import json, sys, re
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n_tests = sys.argv[2] if len(sys.argv) > 2 else "?"
prof_us = sys.argv[3] if len(sys.argv) > 3 else "?"
pw = r["pipeline_weighted"]["config_carturn"]
fs = r["full_step"]
vals = {
    "GPUTESTS": n_tests, "MS": "%.2f" % r["ms_per_step"], "VALUE": "%.1f" % r["value"], "VS": "%.2f" % r["vs_baseline"],
    "LEAN": "%.2f" % r["speedup_vs_torch_gpu_lean_port"], "PWMS": "%.1f" % pw["ms_per_step"], "PWREF": "%.0f" % pw["torch_gpu_ms_per_step"],
    "PWX": "%.0f" % pw["vs_torch_gpu"], "FLASHUS": "%.0f" % r["roofline"]["avg_launch_us"], "FLASHPROF": prof_us,
    "FRAC": "%.3f" % r["roofline"]["frac"], "CFG3": "%.1f" % r["cfg3"]["ms_per_step"],
    "CFG3L": " / ".join("%.2f" % v for v in r["cfg3"]["per_layer_ms"]) + " ms", "CFG3STEP": "%.1f" % r["cfg3_step"]["value"],
    "FULL": "%.1f" % fs["fresco_amd_schedule_mean_ms"], "FULLOPT": "%.0f" % fs["fresco_amd_with_optimisation_ms"],
    "FULLREF": "%.1f" % fs["reference_torch_path_schedule_mean_ms"], "F3ALL": "%.1f" % r["f3_gmflow"]["get_flow_and_interframe_paras_ms"],
    "F3": "%.1f" % r["f3_gmflow"]["gmflow_forward_ms"],
}
s = open("DESIGN.md").read()
for k, v in vals.items():
    s = s.replace("@%s@" % k, v)
left = re.findall(r"@[A-Z0-9]+@", s)
open("DESIGN.md", "w").write(s)
print("filled; left:", left)
