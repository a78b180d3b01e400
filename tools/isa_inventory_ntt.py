#!/usr/bin/env python3
"""Static instruction inventory of the NTT kernels from the assembly the build assembled (kzg_amd/build/ntt_dev_pp.s): per basic
block of k_ntt_pass1 / k_ntt_pass2<false>, instructions by class.  The blocks are then attributed by their shape: the block with
4 x 143 multiply-adds is the radix-4 body of a stage pair (executed once per 4 elements and pair), the one with 143 the first
pair, the ones with global loads / stores the tile load and the epilogue (4 elements per thread).
   python tools/isa_inventory_ntt.py [path/to/ntt_dev_pp.s]  ->  profiles/r05_isa_inventory_ntt.txt"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cls(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if op.startswith(("v_lshrrev_b64", "v_ashrrev_i64", "v_lshlrev_b64")):
        return "shift64"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("v_and_b32", "v_bfe_u32", "v_bfe_i32")):
        return "v_and/bfe (mask)"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add3_u32", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_subrev_co", "v_subbrev")):
        return "v_add/sub 32"
    if op.startswith(("v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_alignbit", "v_lshl_or", "v_lshl_add", "v_and_or", "v_or_b32", "v_or3")):
        return "v_shift/or 32"
    if op.startswith(("v_xor", "v_xad")):
        return "v_xor"
    if op.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_readfirstlane", "v_readlane")):
        return "v_mov/cndmask"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mad_u32")):
        return "v_mul 32"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "VALU other: " + op.split("_e")[0]
    return "other: " + op


def functions(lines):
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN3kzg\w+):", lines[i])
        if m:
            j = i
            while not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[m.group(1)] = lines[i:j]
            i = j
        i += 1
    return out


def blocks_of(F):
    blocks, cur = [], ["entry", []]
    for l in F:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"^\.?[A-Za-z_0-9$]+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], []]
            continue
        if t.startswith("."):
            continue
        cur[1].append(t.split()[0])
    blocks.append(cur)
    return blocks


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    path = args[0] if args else os.path.join(ROOT, "kzg_amd", "build", "ntt_dev_pp.s")
    fs = functions(open(path).read().split("\n"))
    for key, label in (("k_ntt_pass1ILb1", "k_ntt_pass1<DEFER>"), ("k_ntt_pass2ILb0ELb1", "k_ntt_pass2<!SHORT, TWIN>")):
        cand = [n for n in fs if key in n]
        if not cand:    # an assembly from before the templates
            cand = [n for n in fs if key.split("ILb")[0] in n]
        name = cand[0]
        bl = blocks_of(fs[name])
        print("== %s: %d basic blocks, %d instructions" % (label, len(bl), sum(len(b[1]) for b in bl)))
        for n, ins in bl:
            if len(ins) < 12:
                continue
            c = collections.Counter(cls(op) for op in ins)
            valu = sum(v for k, v in c.items() if k not in ("LDS", "VMEM", "s_waitcnt", "s_nop", "s_barrier", "SALU") and not k.startswith("other"))
            print("  block %-12s %5d instr, %5d VALU | " % (n, len(ins), valu) + ", ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))
        print()


if __name__ == "__main__":
    main()
