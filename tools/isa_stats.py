#!/usr/bin/env python
"""Static instruction mix of the pair loop of a kernel, counted from the gfx950 ISA of THIS build.

Why: bench.py prices the dominant kernel two ways that must not exceed 1 by construction --
  executed fp64 flops / time / 78.6 TF   and   issued VALU wave-instructions / time / measured issue ceiling --
and both need "what one rotation step of one wave executes".  That is a property of the compiled loop body, so it
is read off the disassembly (hipcc -S --cuda-device-only of csrc/rmb_sym.hip and rmb_sweep.hip) instead of being replayed from an
old profile.  `SQ_INSTS_VALU` of the rocprofv3 --pmc passes under profiles/ cross-checks the count.

The pair loop = the innermost loop (label ... backward branch) of the kernel with the most fp64 VALU instructions;
blocks the compiler placed out of line behind the backward branch (the near-field r < 2a patch, taken only when
some lane overlaps) are not part of the count.

  python tools/isa_stats.py            -> writes rigidmultiblobswall_amd/librmb_mobility.isa.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rigidmultiblobswall_amd", "csrc")
OUT = os.path.join(ROOT, "rigidmultiblobswall_amd", "librmb_mobility.isa.json")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# name in the JSON -> mangled-name prefix of the kernel
KERNELS = {
    "sym_tt_wall": "_ZN3rmb10sym_kernelILi0ELb1ELb0EEE",
    "sym_coop_tt_wall": "_ZN3rmb15sym_coop_kernelILi0ELb1ELb0EEE",
    "sym2t_tt_wall": "_ZN3rmb12sym2t_kernelILi0ELb1EEE",      # two target blobs per lane: a step evaluates TWO pairs
    "sym_tt_nowall": "_ZN3rmb10sym_kernelILi0ELb0ELb0EEE",
    "sym_tr_wall": "_ZN3rmb10sym_kernelILi1ELb1ELb0EEE",
    "sym_rt_wall": "_ZN3rmb10sym_kernelILi2ELb1ELb0EEE",
    "sym_rr_wall": "_ZN3rmb10sym_kernelILi3ELb1ELb0EEE",
    "sweep_tt_wall": "_ZN3rmb12sweep_kernelILi0ELb1ELb0EEE",
    "sym2_tt_wall": "_ZN3rmb11sym2_kernelILb1ELb0EEE",
    "symx_single_tt_wall": "_ZN3rmb11symx_kernelINS_8OpSingleILi0EEELb1ELb0ELb0EEE",
    "symx_fused_wall": "_ZN3rmb11symx_kernelINS_10OpFusedRowELb1ELb0ELb0EEE",
    "symx_grand_wall": "_ZN3rmb11symx_kernelINS_7OpGrandELb1ELb0ELb0EEE",
    "symx_column_wall": "_ZN3rmb11symx_kernelINS_9OpColumnFELb1ELb0ELb0EEE",
    "symx_tt2_wall": "_ZN3rmb11symx_kernelINS_7OpKindKILi0ELi2EEELb1ELb0ELb0EEE",
    "symx_tt3_wall": "_ZN3rmb11symx_kernelINS_7OpKindKILi0ELi3EEELb1ELb0ELb0EEE",
    "symx_tt4_wall": "_ZN3rmb11symx_kernelINS_7OpKindKILi0ELi4EEELb1ELb0ELb0EEE",
    "symx_rr2_wall": "_ZN3rmb11symx_kernelINS_7OpKindKILi3ELi2EEELb1ELb0ELb0EEE",
    "symx_radii_wall": "_ZN3rmb11symx_kernelINS_9OpRadiiTTELb1ELb0ELb0EEE",
    "symx_free": "_ZN3rmb11symx_kernelINS_13OpFreeSurfaceELb0ELb0ELb0EEE",
}


def source_hash():
  h = hashlib.sha1()
  for f in sorted(os.listdir(CSRC)):
    with open(os.path.join(CSRC, f), "rb") as fh:
      h.update(f.encode()); h.update(fh.read())
  return h.hexdigest()


# the translation units that hold the priced kernels (rmb_internal.h: symmetric family / one-sided family)
ASM_UNITS = ("rmb_sym.hip", "rmb_sweep.hip")


def device_asm():
  from concurrent.futures import ThreadPoolExecutor

  def one(unit):
    return subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S",
                           "--cuda-device-only", "-o", "-", os.path.join(CSRC, unit)],
                          check=True, capture_output=True, text=True).stdout
  with ThreadPoolExecutor(max_workers=len(ASM_UNITS)) as pool:
    return "\n".join(pool.map(one, ASM_UNITS))


def _classify(op):
  """-> (is_valu, flops per lane, class)"""
  if not op.startswith("v_"):
    return False, 0, "other"
  base = op
  for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
    if base.endswith(suf):
      base = base[:-len(suf)]
  if base in ("v_fma_f64", "v_fmac_f64"):
    return True, 2, "fma_f64"
  if base in ("v_mul_f64", "v_add_f64", "v_min_f64", "v_max_f64"):
    return True, 1, "mul_add_f64"
  if base in ("v_rsq_f64", "v_rcp_f64", "v_sqrt_f64"):
    return True, 1, "trans_f64"
  if base.endswith("_f64") or "_f64_" in base:
    return True, 0, "other_f64"          # compares, rounding, ldexp, conversions: issue slots, no flops
  return True, 0, "int_or_move"


def kernel_loop_stats(asm, mangled_prefix):
  """Instruction mix of the pair loop: LLVM annotates every block with the loop it belongs to
  (`; =>This Inner Loop Header` / `; in Loop: Header=BBf_n`); the pair loop is the innermost loop with the most fp64
  VALU instructions.  Member blocks that are the near-field patch (a handful of instructions around v_cndmask or under
  s_and_saveexec, entered
  only when some lane has r < 2a, leaving through s_branch) are left out of the per-step count."""
  lines = asm.split("\n")
  start = None
  for i, l in enumerate(lines):
    if l.startswith(mangled_prefix) and ":" in l:
      start = i
      break
  if start is None:
    return None
  end = start
  while end < len(lines) and "s_endpgm" not in lines[end]:
    end += 1
  body = lines[start:end + 1]
  blocks = []   # (label, annotation, [instruction mnemonics])
  cur = None
  for l in body:
    m = re.match(r"^\.(LBB\d+_\d+):(.*)$", l)
    if m:
      cur = [m.group(1), m.group(2), []]
      blocks.append(cur)
      continue
    if cur is None:
      continue
    if re.match(r"^\s+;", l):
      cur[1] += " " + l.strip()
      continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m:
      cur[2].append(m.group(1))
  best = None
  for label, ann, _ in blocks:
    if "Inner Loop Header" not in ann:
      continue
    hdr = label[1:]           # LBB75_32 -> BB75_32
    members = [b for b in blocks if b[0] == label or re.search(r"in Loop: Header=%s\b" % hdr, b[1])]
    counts = {}
    valu = flops = lds = patch = 0
    for lb, an, ops in members:
      is_patch = (lb != label and len(ops) <= 16 and any(o.startswith(("v_cndmask", "s_and_saveexec")) for o in ops) and ops and
                  ops[-1] == "s_branch")
      if is_patch:
        patch += len(ops)
        continue
      for op in ops:
        is_valu, fl, cls = _classify(op)
        if is_valu:
          valu += 1
          flops += fl
          counts[cls] = counts.get(cls, 0) + 1
        elif op.startswith("ds_"):
          lds += 1
    f64 = sum(v for k, v in counts.items() if k.endswith("f64"))
    # the one-sided sweep has two copies of its pair loop: the bulk one and the one for the single source tile that
    # overlaps the workgroup's own targets, which carries the i == j test (a 64-bit integer compare + exec masking) and
    # a few more instructions.  The bulk loop is the one that is priced.
    self_test = any(op.startswith("v_cmp_ne_u64") or op.startswith("v_cmp_eq_u64") for _, _, ops in members for op in ops)
    if self_test and best is not None and f64 < best["f64_valu_per_step"] + 8:
      continue
    if best is not None and best.get("_self_test") and f64 > best["f64_valu_per_step"] - 8:
      best = None
    if best is None or f64 > best["f64_valu_per_step"]:
      best = {"valu_per_step": valu, "f64_valu_per_step": f64, "flops_per_lane_step": flops, "lds_per_step": lds,
              "classes": counts, "near_field_patch_instructions_excluded": patch, "loop_header": label,
              "_self_test": self_test}
  return best


def generate(path=OUT):
  asm = device_asm()
  res = {"source_sha1": source_hash(), "compiler": subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.split("\n")[0],
         "method": "static count over the innermost pair loop of the device ISA (tools/isa_stats.py); out-of-line near-field patch excluded",
         "kernels": {}}
  for name, pref in KERNELS.items():
    st = kernel_loop_stats(asm, pref)
    if st is not None:
      st.pop("_self_test", None)
      st["pairs_per_step"] = 2 if name.startswith("sym2t_") else 1
      res["kernels"][name] = st
  with open(path, "w") as fh:
    json.dump(res, fh, indent=1)
  return res


def load(regenerate=True):
  """Stats of the current sources: the cached JSON if its hash matches, else regenerated (needs hipcc), else None."""
  try:
    with open(OUT) as fh:
      res = json.load(fh)
    if res.get("source_sha1") == source_hash():
      return res
  except (OSError, ValueError):
    pass
  if regenerate:
    try:
      return generate()
    except (OSError, subprocess.CalledProcessError):
      return None
  return None


if __name__ == "__main__":
  r = generate()
  for k, v in r["kernels"].items():
    print("%-22s VALU/step %4d  (f64 %4d)  flops/lane-step %4d  LDS %2d  %s" % (k, v["valu_per_step"], v["f64_valu_per_step"],
                                                                                 v["flops_per_lane_step"], v["lds_per_step"], v["classes"]))
