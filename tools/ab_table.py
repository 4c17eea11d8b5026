"""Ratio table of a same-box A/B of two builds timed by tools/bench_ops.py (RMB_AB_LIB / RMB_AB_OUT):
  python tools/ab_table.py gpurun_out/ab_base1.json gpurun_out/ab_base2.json gpurun_out/ab_new1.json gpurun_out/ab_new2.json"""
import json
import sys

L = lambda f: {(r["N"], r["product"], r["path"]): r["kernel_ms"] for r in json.load(open(f))}
b1, b2, n1, n2 = (L(f) for f in sys.argv[1:5])
print("%-7s %-34s %-44s %9s %9s %9s %9s %6s" % ("N", "product", "path", "base#1", "base#2", "new#1", "new#2", "ratio"))
for k in b1:
  b, n = (b1[k] + b2[k]) / 2, (n1[k] + n2[k]) / 2
  print("%-7d %-34s %-44s %9.4f %9.4f %9.4f %9.4f %6.3f" % (k[0], k[1][:34], k[2][:44], b1[k], b2[k], n1[k], n2[k], b / n))
