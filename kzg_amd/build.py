"""Builds libkzg_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch involved.

msm.hip (the bucket-accumulation kernel) and ntt.hip go through their assembly: hipcc -S for the device side, tools-free
post-processing (strip_asm_nops below), then assembler, lld and the offload bundler exactly as hipcc itself would run them, and the
host side compiled against that device image.  Everything else is a plain `hipcc -c`; so are those two when the compiler is not the
one the post-processing was validated with (asm_path_ok)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkzg_mi355x.so")
# the same library with the unit-test hooks of include/kzg_mi355x_test.h compiled in (-DKZG_TEST_HOOKS): loaded by tests/ only
OUT_HOOKS = os.path.join(HERE, "libkzg_mi355x_hooks.so")
HOOK_SOURCES = ["capi.hip", "mgpu.hip"]  # the translation units that hold hooks (runtime.hip holds none)
SOURCES = ["capi.hip", "runtime.hip", "msm.hip", "srs.hip", "ntt.hip", "poly.hip", "witness.hip", "pairing.hip", "msm_wide.hip", "msm_tail.hip", "mgpu.hip", "gfft.hip"]
# per-file extra flags (none at present; out-of-line multiplies for the tail kernels were measured: no gain)
EXTRA_FLAGS = {}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
# translation units whose device assembly is post-processed
VIA_ASM = ["msm.hip", "ntt.hip"]
LLVM_BIN = os.environ.get("KZG_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
# the compiler the post-processing rule was validated with (its own nop placement and hazard model); any other compiler, or a
# missing assembler / linker / bundler, gets the plain `hipcc -c` path with the nops in place
LLVM_VALIDATED = "roc-7.2.0"
_asm_path_ok = None


def asm_path_ok(hipcc):
    global _asm_path_ok
    if _asm_path_ok is None:
        try:
            ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        except OSError:
            ver = ""
        tools = all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("clang", "lld", "clang-offload-bundler"))
        _asm_path_ok = (LLVM_VALIDATED in ver) and tools
        if not _asm_path_ok:
            sys.stderr.write("kzg_amd.build: compiler is not the validated %s (or LLVM tools missing under %s): msm.hip / ntt.hip are "
                             "built without the s_nop post-processing\n" % (LLVM_VALIDATED, LLVM_BIN))
    return _asm_path_ok


def strip_asm_nops(text):
    """hipcc puts `s_nop 0` behind every inline-asm block whose result the next instruction reads: on gfx950 it must assume the
    block ended in an instruction with the dst_sel / cvt-scale forwarding hazard (one wait state).  The blocks of
    mul30_gfx950.inc end in v_mad_i64_i32, which has no such hazard (the compiler itself follows its own v_mad_i64_i32 with a
    dependent v_ashrrev_i64 without a wait state), so those 250 nops per bucket addition -- each costs the wave an issue
    slot, measured 0.6 of a multiply-add at two waves per SIMD (profiles/r03_issue_cost.txt) -- are removed.  Only a nop that
    directly follows a block whose last instruction is v_mad_i64_i32 / v_mad_u64_u32 goes, and only when the NEXT instruction is a
    plain VALU / SALU / LDS / memory instruction: in front of anything that reads VGPRs across lanes or with a sub-dword selector
    (v_readlane / v_readfirstlane / v_writelane / v_permlane*, DPP and SDWA forms) the nop stays -- there the wait state could be a
    real hazard of another kind.  Returns (text, removed).  Validated for the hipcc of ROCm 7.2 (LLVM_VALIDATED below):
    with another compiler the build keeps the nops and says so."""
    lines = text.split("\n")
    out, removed = [], 0
    last_asm_insn = None       # last instruction line seen inside the current / most recent asm block
    in_asm = False
    just_ended = False

    def next_insn(k):
        for m in range(k + 1, min(k + 40, len(lines))):
            u = lines[m].strip()
            if u and not u.startswith((";", ".")) and not u.endswith(":"):
                return u
        return ""

    def crosses_lanes(u):
        if not u:            # no instruction follows within reach (end of a function, directives): nothing is known, the nop stays
            return True
        return u.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_mov_b32_dpp", "ds_swizzle", "ds_bpermute",
                             "ds_permute")) or "_dpp" in u.split()[0] or "_sdwa" in u.split()[0] or " dpp" in u or "quad_perm" in u or "row_" in u

    for k, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, last_asm_insn, just_ended = True, None, False
        elif t.startswith(";;#ASMEND"):
            in_asm, just_ended = False, True
        elif in_asm:
            if t and not t.startswith(";"):
                last_asm_insn = t
        else:
            if (just_ended and re.match(r"s_nop\s+0\s*$", t) and last_asm_insn and last_asm_insn.startswith(("v_mad_i64_i32", "v_mad_u64_u32"))
                    and not crosses_lanes(next_insn(k))):
                removed += 1
                just_ended = False
                continue
            if t and not t.startswith(";"):
                just_ended = False
        out.append(ln)
    return "\n".join(out), removed


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, defines=(), strip_nops=True, tag=None):
    """out / defines / strip_nops / tag: A/B variants (tools/ab_build.py): objects go to build/<tag>/, no hooks library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # every header / generated include: a stale object after a header-only edit would silently ship old kernels
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "kzg_mi355x.h"))
    hdrs.append(os.path.join(HERE, "..", "include", "kzg_mi355x_test.h"))
    hdrs.append(os.path.abspath(__file__))
    objdir = os.path.join(HERE, "build", tag) if tag else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + ["-D" + d for d in defines]
    variant = tag is not None
    target = out or OUT

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("build step failed: " + " ".join(cmd))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)

    def via_asm(src, obj, extra):
        base = obj[:-2]
        run([hipcc] + flags + extra + ["--cuda-device-only", "-S", src, "-o", base + "_dev.s"])
        text = open(base + "_dev.s").read()
        removed = 0
        if strip_nops:
            text, removed = strip_asm_nops(text)
        open(base + "_dev_pp.s", "w").write(text)
        if verbose:
            print(f"{os.path.basename(src)}: {removed} s_nop removed behind v_mad blocks", flush=True)
        run([os.path.join(LLVM_BIN, "clang"), "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + "_dev_pp.s", "-o", base + "_dev.o"])
        run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + "_dev.hsaco",
             base + "_dev.o"])
        run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + base + "_dev.hsaco",
             "-output=" + base + ".hipfb"])
        run([hipcc] + flags + extra + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-c", src, "-o", obj])

    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        extra = EXTRA_FLAGS.get(s, [])
        if force or _stale(obj, [src] + hdrs):
            if s in VIA_ASM and asm_path_ok(hipcc):
                jobs.append(("asm", src, obj, extra))
            else:
                jobs.append(("cc", [hipcc] + flags + extra + ["-c", src, "-o", obj]))
        if s in HOOK_SOURCES and not variant:
            hobj = os.path.join(objdir, s.replace(".hip", "_hooks.o"))
            if force or _stale(hobj, [src] + hdrs):
                jobs.append(("cc", [hipcc] + flags + extra + ["-DKZG_TEST_HOOKS", "-c", src, "-o", hobj]))

    def do(job):
        if job[0] == "asm":
            via_asm(job[1], job[2], job[3])
        else:
            run(job[1])

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(do, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(target, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-ldl", "-lpthread"])
    if not variant:
        hobjs = [os.path.join(objdir, s.replace(".hip", "_hooks.o" if s in HOOK_SOURCES else ".o")) for s in srcs]
        if force or jobs or _stale(OUT_HOOKS, hobjs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT_HOOKS] + hobjs + ["-ldl", "-lpthread"])
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
