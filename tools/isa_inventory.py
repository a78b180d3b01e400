#!/usr/bin/env python3
"""Static instruction inventory of k_accum_affine's hot path from the compiler's assembly.
   python tools/isa_inventory.py   (compiles kzg_amd/csrc/msm.hip with --save-temps into a temp dir)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cls(op):
    if op.startswith(("v_mad_i64_i32", "v_mad_u64_u32")):
        return "v_mad_i64_i32 (multiply-add)"
    if op.startswith("s_nop"):
        return "s_nop (after every inline-asm block; not VALU)"
    if op.startswith("s_"):
        return "other scalar (not VALU)"
    if op.startswith("global_"):
        return "vmem (not VALU)"
    if op.startswith(("v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64")):
        return "64-bit shift (carry to the next column)"
    if op.startswith("v_lshl_add_u64"):
        return "v_lshl_add_u64 (+2^29 rounding before the digit split)"
    if op.startswith("v_mul_lo"):
        return "v_mul_lo_u32 (Montgomery quotient digit)"
    if op.startswith("v_bfe_i32"):
        return "v_bfe_i32 (balanced digit extraction)"
    if op.startswith("v_ashrrev_i32"):
        return "v_ashrrev_i32 (sign-extend quotient digit / normalize carries)"
    return "other 32-bit VALU: " + op.split("_e")[0]


def main():
    """python tools/isa_inventory.py [path/to/msm_dev_pp.s]: default = the assembly the build actually assembled
    (kzg_amd/build/msm_dev_pp.s, after strip_asm_nops); `--compile` recompiles msm.hip the plain way instead."""
    if "--compile" in sys.argv:
        tmp = tempfile.mkdtemp()
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c",
                               os.path.join(ROOT, "kzg_amd", "csrc", "msm.hip"), "-o", os.path.join(tmp, "msm.o"), "--save-temps"],
                              cwd=tmp, stderr=subprocess.DEVNULL)
        path = os.path.join(tmp, "msm-hip-amdgcn-amd-amdhsa-gfx950.s")
    else:
        args = [a for a in sys.argv[1:] if not a.startswith("-")]
        path = args[0] if args else os.path.join(ROOT, "kzg_amd", "build", "msm_dev_pp.s")
    L = open(path).read().split("\n")
    start = [i for i, l in enumerate(L) if l.startswith("_ZN3kzg14k_accum_affine")][0]
    end = [i for i, l in enumerate(L) if l.startswith(".Lfunc_end") and i > start][0]
    blocks, cur = [], ("entry", [])
    for l in L[start:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"^\.?[A-Za-z_0-9$]+:", t):
            blocks.append(cur)
            cur = (t.split(":")[0], [])
            continue
        if t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if t:
            cur[1].append(t)
    blocks.append(cur)
    # the mixed addition = the two blocks of the loop body with the most multiply-adds (phase 1: two products, phase 2: the rest)
    mads = lambda b: sum(i.startswith("v_mad_i64_i32") for i in b[1])  # noqa: E731
    # phase 1 = two products (676 multiply-adds, 702 with the merged subtractions), phase 2 = the rest (2379 / 2405); the other
    # large block of the loop is the doubling branch (never taken on the hot path)
    hot = [b for b in blocks if mads(b) in (676, 702, 2379, 2405)]
    c = collections.Counter()
    for _, ins in hot:
        for i in ins:
            c[cls(i.split()[0])] += 1
    n = sum(c.values())
    valu = sum(v for k, v in c.items() if "not VALU" not in k)
    print("%s\nk_accum_affine, one XYZZ mixed addition (blocks %s with %s multiply-adds): %d instructions, %d of them VALU" %
          (os.path.relpath(path, ROOT), ", ".join(b[0] for b in hot), " + ".join(str(mads(b)) for b in hot), n, valu))
    for k, v in c.most_common():
        print("%6d  %5.1f %%  %s" % (v, 100.0 * v / n, k))


if __name__ == "__main__":
    main()
