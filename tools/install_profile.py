"""Copy the summaries of one tools/profile_bench.sh run into profiles/ and record its PMC figures in profiles/traffic.json
(the entry bench.py's `roofline.traffic` cites).

  python tools/install_profile.py gpurun_out/prof_<tag> <N> <label> [prefix]
    e.g.  python tools/install_profile.py gpurun_out/prof_r2_final_N1e4 10000 N1e4 r2_final

An existing entry <prefix>_sym_tt_wall_N<N> is kept under a `superseded_` key (bench.py takes the last key that ends
in sym_tt_wall_N<N>).
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, N, label = sys.argv[1], int(sys.argv[2]), sys.argv[3]
prefix = sys.argv[4] if len(sys.argv) > 4 else "r2_final"
base = os.path.join(ROOT, "profiles", "%s_bench_%s" % (prefix, label))
shutil.copy(os.path.join(src, "summary.txt"), base + "_rocprofv3_summary.txt")
shutil.copy(os.path.join(src, "summary.json"), base + "_rocprofv3_summary.json")
shutil.copy(os.path.join(src, "bench_line_under_trace.json"), base + "_line_under_rocprofv3_trace.json")

with open(os.path.join(src, "summary.json")) as fh:
  summ = json.load(fh)
kern = [v for k, v in summ.items() if "sym_kernel<0, true, false>" in k or "sym_coop_kernel<0, true, false>" in k or "sym2t_kernel<0, true>" in k]
if not kern:
  raise SystemExit("no rmb::sym_kernel / sym_coop_kernel / sym2t_kernel <TT, wall> in %s/summary.json" % src)
kern.sort(key=lambda v: -v.get("trace_n", 0))     # the one the timed steps ran
k = kern[0]
ub = [v for kk, v in summ.items() if "ubench_fma64_kernel" in kk]
with open(os.path.join(src, "bench_line_under_trace.json")) as fh:
  line = json.loads(fh.read().strip().splitlines()[-1])
entry = {
    "kernel": [kk for kk, v in summ.items() if v is k][0],
    "FETCH_SIZE_kb": k.get("FETCH_SIZE"), "WRITE_SIZE_kb": k.get("WRITE_SIZE"),
    "traffic_bytes": int(round(k["hbm_traffic_bytes_per_launch"])),
    "SQ_INSTS_VALU_per_launch": k.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU_per_launch": k.get("SQ_ACTIVE_INST_VALU"),
    "SQ_BUSY_CYCLES_per_launch": k.get("SQ_BUSY_CYCLES"),
    "valu_active_per_busy_cycle": round(k["SQ_ACTIVE_INST_VALU"] / k["SQ_BUSY_CYCLES"], 3),
    "ubench_valu_active_per_busy_cycle": round(ub[0]["SQ_ACTIVE_INST_VALU"] / ub[0]["SQ_BUSY_CYCLES"], 3) if ub else None,
    "kernel_trace_avg_us_timed_steps": k.get("trace_last_avg_us", k.get("trace_avg_us")),
    "kernel_trace_avg_us_all": k.get("trace_avg_us"),
    "hip_event_kernel_ms_avg_in_the_traced_process": line["roofline"]["kernel_ms_avg"],
    "command": open(os.path.join(src, "command.txt")).read().strip(),
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
    "files": "profiles/%s_bench_%s_rocprofv3_summary.{txt,json}" % (prefix, label),
}
tpath = os.path.join(ROOT, "profiles", "traffic.json")
with open(tpath) as fh:
  tj = json.load(fh)
key = "%s_sym_tt_wall_N%d" % (prefix, N)
out = {}
for kk, v in tj.items():
  if kk == key:
    out["superseded_%s_commit_%s" % (kk.replace("sym_tt_wall_N", "sym_tt_wall_n"), v.get("commit", "x"))] = v
  else:
    out[kk] = v
out[key] = entry
with open(tpath, "w") as fh:
  json.dump(out, fh, indent=1)
print(key, json.dumps(entry, indent=1))
