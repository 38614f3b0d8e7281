#!/usr/bin/env python3
"""VALU issue bound of k4_glcm_thread<7,3> and k4_glcm_pair from (a) the static instruction histogram of the kernel
(profiles/<round>_glcm_*_valu_hist.txt: disassembly of the gfx950 code object, `grep v_ | uniq -c`) and
(b) the measured issue cost of each instruction class at 4 waves per SIMD (profiles/r02_ubench.json, produced by
profiles/ubench/ubench.hip on an MI355X).  Writes profiles/<round>_valu_issue.json (bench.py reads the newest) for the
roofline entry of the texture kernel:  frac = waves * issue_cycles_per_wave / (1024 SIMDs * 2.4 GHz * measured time).

Measured classes (cycles per wave64 instruction on one SIMD, 4 waves per SIMD, independent streams):
  fast  ~1.96   v_add_u32 / v_sub / v_and / v_or / v_xor / v_mov / v_fma_f32 (either encoding)
  slow  ~3.24   shifts, v_pk_* (packed 16-bit), v_perm, v_sad_u8, v_dot*, v_add3, v_or3, v_lshl_add, v_bfe, v_mad*,
                v_mul_lo, conversions and every float64 instruction
An instruction that was not measured is priced as slow."""
import json
import os
import re

import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"   # histogram files <ROUND>_glcm_*_valu_hist.txt -> <ROUND>_valu_issue.json
ub = json.load(open(os.path.join(HERE, "r02_ubench.json")))["valu"]
fast = min(v["4_waves_per_simd"] for k, v in ub.items() if k in ("v_add_u32", "v_and_b32", "v_xor_b32"))
slow = max(v["4_waves_per_simd"] for k, v in ub.items() if k in ("v_pk_min_u16", "v_perm_b32", "v_fma_f64", "v_lshlrev_b32"))
FAST = re.compile(r"^v_(add_u32|add_co_u32|addc_co_u32|sub_u32|sub_co_u32|subb_co_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|mov_b32|mov_b64|fma_f32|add_f32|mul_f32|max_f32|min_f32)(_e32|_e64)?$")


def price(hist_file, windows_per_thread):
    hist = []
    for line in open(os.path.join(HERE, hist_file)):
        n, op = line.split()
        hist.append((op, int(n)))
    n_fast = sum(n for op, n in hist if FAST.match(op))
    n_slow = sum(n for op, n in hist if not FAST.match(op))
    cycles = n_fast * fast + n_slow * slow
    return {"valu_static": n_fast + n_slow, "fast": n_fast, "slow": n_slow, "windows_per_thread": windows_per_thread,
            "issue_cycles_per_wave": round(cycles, 1), "issue_cycles_per_64_windows": round(cycles / windows_per_thread, 1),
            "valu_per_window": round((n_fast + n_slow) / windows_per_thread, 1),
            "packed16_insts": sum(n for op, n in hist if op.startswith("v_pk_")),
            "f64_insts": sum(n for op, n in hist if "f64" in op)}


out = {"source": "profiles/ubench/ubench.hip on MI355X + static histogram of the gfx950 code object",
       "cost_fast_cycles": fast, "cost_slow_cycles": slow,
       # one window per thread: every geometry but the dense one (and the dense one until r02's last day)
       "glcm_thread_7_3": price(f"{ROUND}_glcm_thread_7_3_valu_hist.txt", 1),
       # window 7, step 1, levels <= 32: two adjacent windows per thread share the sort of their common keys
       "glcm_pair": price(f"{ROUND}_glcm_pair_valu_hist.txt", 2)}
if os.path.exists(os.path.join(HERE, f"{ROUND}_glcm_quad_valu_hist.txt")):
    # r04: a 2 x 2 block of windows per thread (histogram of the kernel compiled with its per-window finish loop unrolled, so
    # that static counts are executed counts: profiles/valu_hist.sh)
    out["glcm_quad"] = price(f"{ROUND}_glcm_quad_valu_hist.txt", 4)
json.dump(out, open(os.path.join(HERE, f"{ROUND}_valu_issue.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
