#!/usr/bin/env python3
"""Generates kzg_amd/csrc/mul30_gfx950.inc: the signed 13 x 30-bit Montgomery multiply / square / fused multiply-add of
field30.h with every column's products in ONE v_mad_i64_i32 accumulator chain (inline-asm blocks of up to 30 operands).

hipcc's own code for the portable C version sums each column's products in a second accumulator and merges it with the
carry by a 64-bit add (one v_lshl_add_u64 per column, 36 per multiply, for latency that two resident waves already hide);
here the chain starts from the carry.  Everything that is not a multiply-add (quotient digit, shifts, digit extraction)
stays in C between the blocks."""
import sys

import os

N = 13
MAX_OPS = 30
# experiment: carry-out destination of the multiply-adds rotating over ROT scratch SGPR pairs instead of always vcc (a lone wave
# then issues dependent-free multiply-adds faster: tools/mad_issue.hip); 0 = vcc
ROT = int(os.environ.get("KZG_GEN_ROT", "0"))


def block(products, mnem="v_mad_i64_i32"):
    """products: list of (x_expr, x_cons, y_expr, y_cons); one asm statement accumulating all of them into acc"""
    ops, lines = [], []

    def ref(cons, expr):
        for i, (c, e) in enumerate(ops):
            if e == expr and c == cons:
                return i + 1
        ops.append((cons, expr))
        return len(ops)
    for n_, (x, xc, y, yc) in enumerate(products):
        xtxt = f"%{ref(xc, x) + ROT}"
        ytxt = y if yc == "i" else f"%{ref(yc, y) + ROT}"       # "i": an inline constant written into the instruction
        sdst = f"%{1 + n_ % ROT}" if ROT else "vcc"
        lines.append(f"{mnem} %0, {sdst}, {xtxt}, {ytxt}, %0")
    assert len(ops) + 1 + ROT <= MAX_OPS, len(ops)
    body = "\\n\\t".join(lines)
    ins = ", ".join(f'"{c}"({e})' for c, e in ops)
    if ROT:
        outs = "".join(f', "=&s"(rot_sink[{k}])' for k in range(ROT))
        return f'    {{ uint64_t rot_sink[{ROT}]; asm("{body}" : "+v"(acc){outs} : {ins}); }}\n'
    return f'    asm("{body}" : "+v"(acc) : {ins} : "vcc");\n'


def emit_products(products, mnem="v_mad_i64_i32"):
    """split a column's product list into blocks that respect the operand limit"""
    out, cur, names = "", [], set()
    for p in products:
        new = {(p[1], p[0])} | ({(p[3], p[2])} if p[3] != "i" else set())
        if len(names | new) + 1 + ROT > MAX_OPS:
            out += block(cur, mnem)
            cur, names = [], set()
        cur.append(p)
        names |= new
    if cur:
        out += block(cur, mnem)
    return out


def q(j):
    return f"Fq30Consts::mod({j})"


def digit_out(k, unsigned_out):
    """the output digit of column k and the carry into the next one: balanced (round to nearest) or unsigned (floor: no rounding add)"""
    if unsigned_out:
        return f"    r.v[{k - N}] = (int32_t)((uint32_t)acc & F30_MASK);\n    acc = sar30(acc);\n"
    return f"    r.v[{k - N}] = sext30((uint32_t)acc);\n    acc = sar30(acc + (uint64_t)F30_HALF);\n"


def gen_mul(name="mul30_asm", subs=(), unsigned_out=False):
    """subs: ((operand, constant), ...): the result is a*b/R + sum constant * operand, the extra terms entering the output
    columns as one v_mad_i64_i32 by an inline constant each (a subtraction merged into the product: no separate limb-wise
    subtraction and carry pass, and the result comes out normalised)."""
    args = "".join(f", const Fq30 &{o}" for o, _ in subs)
    s = f"__device__ __forceinline__ Fq30 {name}(const Fq30 &a, const Fq30 &b{args}) {{\n    int32_t m[F30_N];\n    Fq30 r;\n    uint64_t acc = 0;\n"
    for k in range(N):
        prods = [(f"a.v[{i}]", "v", f"b.v[{k - i}]", "v") for i in range(k + 1)]
        prods += [(f"m[{i}]", "v", q(k - i), "s") for i in range(k)]
        s += emit_products(prods)
        s += f"    m[{k}] = sext30((uint32_t)acc * Fq30Consts::INV);\n    acc = sar30(mac30(acc, m[{k}], {q(0)}));\n"
    for k in range(N, 2 * N - 1):
        prods = []
        for i in range(k - N + 1, N):
            prods.append((f"a.v[{i}]", "v", f"b.v[{k - i}]", "v"))
            prods.append((f"m[{i}]", "v", q(k - i), "s"))
        prods += [(f"{o}.v[{k - N}]", "v", str(c), "i") for o, c in subs]
        s += emit_products(prods)
        s += digit_out(k, unsigned_out)
    if subs:
        s += emit_products([(f"{o}.v[{N - 1}]", "v", str(c), "i") for o, c in subs])
    s += f"    r.v[{N - 1}] = (int32_t)acc;\n    return r;\n}}\n"
    return s


def gen_sqr(name="sqr30_asm", subs=(), unsigned_out=False):
    args = "".join(f", const Fq30 &{o}" for o, _ in subs)
    s = (f"__device__ __forceinline__ Fq30 {name}(const Fq30 &a{args}) {{\n    int32_t m[F30_N], d[F30_N];\n    Fq30 r;\n"
         "#pragma unroll\n    for (int i = 0; i < F30_N; i++) d[i] = a.v[i] * 2;\n    uint64_t acc = 0;\n")

    def cross(k, lo):
        p = [(f"a.v[{i}]", "v", f"d[{k - i}]", "v") for i in range(lo, N) if 2 * i < k and k - i < N]
        if k % 2 == 0:
            p.append((f"a.v[{k // 2}]", "v", f"a.v[{k // 2}]", "v"))
        return p
    for k in range(N):
        prods = cross(k, 0) + [(f"m[{i}]", "v", q(k - i), "s") for i in range(k)]
        s += emit_products(prods)
        s += f"    m[{k}] = sext30((uint32_t)acc * Fq30Consts::INV);\n    acc = sar30(mac30(acc, m[{k}], {q(0)}));\n"
    for k in range(N, 2 * N - 1):
        prods = cross(k, k - N + 1) + [(f"m[{i}]", "v", q(k - i), "s") for i in range(k - N + 1, N)]
        prods += [(f"{o}.v[{k - N}]", "v", str(c), "i") for o, c in subs]
        s += emit_products(prods)
        s += digit_out(k, unsigned_out)
    if subs:
        s += emit_products([(f"{o}.v[{N - 1}]", "v", str(c), "i") for o, c in subs])
    s += f"    r.v[{N - 1}] = (int32_t)acc;\n    return r;\n}}\n"
    return s


def gen_muladd():
    """(a*b + c*d)/R30 with one reduction; in the columns with more than 30 products the multiple of 2^30 accumulated so far is
    set aside before the c*d products go in (field30.h, muladd30_inline)."""
    s = ("__device__ __forceinline__ Fq30 muladd30_asm(const Fq30 &a, const Fq30 &b, const Fq30 &c, const Fq30 &d) {\n"
         "    int32_t m[F30_N];\n    Fq30 r;\n    uint64_t acc = 0, hi;\n")
    for k in range(2 * N - 1):
        lo = max(0, k - N + 1)
        hi_i = min(k, N - 1)
        cnt = hi_i - lo + 1
        split = 3 * cnt > 30
        ab = [(f"a.v[{i}]", "v", f"b.v[{k - i}]", "v") for i in range(lo, hi_i + 1)]
        cd = [(f"c.v[{i}]", "v", f"d.v[{k - i}]", "v") for i in range(lo, hi_i + 1)]
        mq = [(f"m[{i}]", "v", q(k - i), "s") for i in range(lo, min(k, N)) if k - i >= 1 or k >= N]
        if k < N:
            mq = [(f"m[{i}]", "v", q(k - i), "s") for i in range(k)]
        else:
            mq = [(f"m[{i}]", "v", q(k - i), "s") for i in range(k - N + 1, N)]
        if split:
            s += emit_products(ab + mq)
            s += "    hi = sar30(acc);\n    acc &= (uint64_t)F30_MASK;\n"
            s += emit_products(cd)
        else:
            s += emit_products(ab + mq + cd)
        tail = " + hi" if split else ""
        if k < N:
            s += f"    m[{k}] = sext30((uint32_t)acc * Fq30Consts::INV);\n    acc = sar30(mac30(acc, m[{k}], {q(0)})){tail};\n"
        else:
            s += f"    r.v[{k - N}] = sext30((uint32_t)acc);\n    acc = sar30(acc + (uint64_t)F30_HALF){tail};\n"
    s += f"    r.v[{N - 1}] = (int32_t)acc;\n    return r;\n}}\n"
    return s


def gen_fr29():
    """Fr in 9 unsigned 29-bit limbs (fr29.h, the NTT kernels): same single-chain structure with v_mad_u64_u32."""
    n = 9

    def blocku(products):
        return emit_products(products, "v_mad_u64_u32")

    def qr(j):
        return f"Fr29Consts::mod({j})"
    s = "__device__ __forceinline__ Fr29 mul29r_asm(const Fr29 &a, const Fr29 &b) {\n    uint32_t m[R29_N];\n    Fr29 r;\n    uint64_t acc = 0;\n"
    for k in range(n):
        prods = [(f"a.v[{i}]", "v", f"b.v[{k - i}]", "v") for i in range(k + 1)]
        prods += [(f"m[{i}]", "v", qr(k - i), "s") for i in range(k)]
        s += blocku(prods)
        s += f"    m[{k}] = ((uint32_t)acc * Fr29Consts::INV) & F29_MASK;\n    acc = (acc + (uint64_t)m[{k}] * {qr(0)}) >> 29;\n"
    for k in range(n, 2 * n - 1):
        prods = []
        for i in range(k - n + 1, n):
            prods.append((f"a.v[{i}]", "v", f"b.v[{k - i}]", "v"))
            prods.append((f"m[{i}]", "v", qr(k - i), "s"))
        s += blocku(prods)
        s += f"    r.v[{k - n}] = (uint32_t)acc & F29_MASK;\n    acc >>= 29;\n"
    s += f"    r.v[{n - 1}] = (uint32_t)acc;\n    return r;\n}}\n"
    return s


def gen_shoup29():
    """Fr in 9 unsigned 29-bit limbs, multiplication by a CONSTANT w given with wp = floor(w 2^261 / r) (Shoup / Harvey): the
    quotient comes from the high columns of x * wp (columns 7..16: 53 products, column 7 as the guard), the result from the low nine
    columns of x * w - q * r in one signed chain (45 + 45 products, r_0 = 1 enters as the inline constant -1): 143 multiply-adds
    and 37 other instructions against the 154 + 9 + 53 of the Montgomery product (mul29r_asm), no quotient-digit multiplies."""
    n = 9
    s = ("__device__ __forceinline__ Fr29 mulshoup29_asm(const Fr29 &x, const Fr29 &w, const Fr29 &wp) {\n    uint32_t q[R29_N];\n    Fr29 r;\n"
         "    uint64_t acc = 0;\n")
    for k in range(7, 2 * n - 1):
        prods = [(f"x.v[{i}]", "v", f"wp.v[{k - i}]", "v") for i in range(max(0, k - n + 1), min(n - 1, k) + 1)]
        s += emit_products(prods, "v_mad_u64_u32")
        if k >= n:
            s += f"    q[{k - n}] = (uint32_t)acc & F29_MASK;\n"
        s += "    acc >>= 29;\n"
    s += f"    q[{n - 1}] = (uint32_t)acc;\n    acc = 0;\n"
    for k in range(n):
        prods = [(f"x.v[{i}]", "v", f"w.v[{k - i}]", "v") for i in range(k + 1)]
        for i in range(k + 1):
            j = k - i
            prods.append((f"q[{i}]", "v", "-1", "i") if j == 0 else (f"q[{i}]", "v", f"(int32_t)(0u - Fr29Consts::mod({j}))", "s"))
        s += emit_products(prods, "v_mad_i64_i32")
        s += f"    r.v[{k}] = (uint32_t)acc & F29_MASK;\n"
        if k < n - 1:
            s += "    acc = (uint64_t)((int64_t)acc >> 29);\n"
    s += "    return r;\n}\n"
    return s


def block2(products, mnem):
    """two interleaved accumulator chains in one asm statement: products = list of (chain, x, xc, y, yc); chain 0 -> acc (%0), chain 1 -> acc2 (%1)"""
    ops, lines = [], []

    def ref(cons, expr):
        for i, (c, e) in enumerate(ops):
            if e == expr and c == cons:
                return i + 2
        ops.append((cons, expr))
        return len(ops) + 1
    for ch, x, xc, y, yc in products:
        xtxt = f"%{ref(xc, x)}"
        ytxt = y if yc == "i" else f"%{ref(yc, y)}"
        lines.append(f"{mnem} %{ch}, vcc, {xtxt}, {ytxt}, %{ch}")
    assert len(ops) + 2 <= MAX_OPS, len(ops)
    body = "\\n\\t".join(lines)
    ins = ", ".join(f'"{c}"({e})' for c, e in ops)
    return f'    asm("{body}" : "+v"(acc), "+v"(acc2) : {ins} : "vcc");\n'


def emit_products2(pa, pb, mnem):
    """columns of two independent products, multiply-add by multiply-add alternating between the two chains; split into blocks that
    respect the operand limit"""
    seq = []
    for i in range(max(len(pa), len(pb))):
        if i < len(pa):
            seq.append((0,) + pa[i])
        if i < len(pb):
            seq.append((1,) + pb[i])
    out, cur, names = "", [], set()
    for pr in seq:
        new = {(pr[2], pr[1])} | ({(pr[4], pr[3])} if pr[4] != "i" else set())
        if len(names | new) + 2 > MAX_OPS:
            out += block2(cur, mnem)
            cur, names = [], set()
        cur.append(pr)
        names |= new
    if cur:
        out += block2(cur, mnem)
    return out


def gen_shoup29_dual():
    """Two Shoup products (x * w, y * w2) as ONE instruction stream with the two accumulator chains interleaved multiply-add by
    multiply-add: a wave then has two independent dependency chains in flight (a dependent v_mad_u64_u32 issues every 5.3 cycles, an
    independent one every 4.3: profiles/r03_issue_cost.txt) -- for the NTT's stage pairs, whose four products per butterfly come in
    two independent pairs."""
    n = 9
    s = ("__device__ __forceinline__ void mulshoup29x2_asm(const Fr29 &x, const Fr29 &w, const Fr29 &wp, const Fr29 &y, const Fr29 &w2, const Fr29 &wp2,\n"
         "                                                 Fr29 &rx, Fr29 &ry) {\n    uint32_t q[R29_N], q2[R29_N];\n    Fr29 r, r2;\n"
         "    uint64_t acc = 0, acc2 = 0;\n")
    for k in range(7, 2 * n - 1):
        rng = range(max(0, k - n + 1), min(n - 1, k) + 1)
        pa = [(f"x.v[{i}]", "v", f"wp.v[{k - i}]", "v") for i in rng]
        pb = [(f"y.v[{i}]", "v", f"wp2.v[{k - i}]", "v") for i in rng]
        s += emit_products2(pa, pb, "v_mad_u64_u32")
        if k >= n:
            s += f"    q[{k - n}] = (uint32_t)acc & F29_MASK;\n    q2[{k - n}] = (uint32_t)acc2 & F29_MASK;\n"
        s += "    acc >>= 29;\n    acc2 >>= 29;\n"
    s += f"    q[{n - 1}] = (uint32_t)acc;\n    q2[{n - 1}] = (uint32_t)acc2;\n    acc = 0;\n    acc2 = 0;\n"
    for k in range(n):
        def col(xn, wn, qn):
            prods = [(f"{xn}.v[{i}]", "v", f"{wn}.v[{k - i}]", "v") for i in range(k + 1)]
            for i in range(k + 1):
                j = k - i
                prods.append((f"{qn}[{i}]", "v", "-1", "i") if j == 0 else (f"{qn}[{i}]", "v", f"(int32_t)(0u - Fr29Consts::mod({j}))", "s"))
            return prods
        s += emit_products2(col("x", "w", "q"), col("y", "w2", "q2"), "v_mad_i64_i32")
        s += f"    r.v[{k}] = (uint32_t)acc & F29_MASK;\n    r2.v[{k}] = (uint32_t)acc2 & F29_MASK;\n"
        if k < n - 1:
            s += "    acc = (uint64_t)((int64_t)acc >> 29);\n    acc2 = (uint64_t)((int64_t)acc2 >> 29);\n"
    s += "    rx = r;\n    ry = r2;\n}\n"
    return s


def main(dst):
    out = ("// GENERATED by tools/gen_mul30.py -- do not edit.\n" + gen_mul() + gen_sqr() + gen_muladd() +
           gen_mul("mul30_sub_asm", (("c", -1),)) + gen_sqr("sqr30_sub2_asm", (("c", -1), ("e", -2))) +
           gen_mul("mul30u_asm", unsigned_out=True) + gen_sqr("sqr30_sub2u_asm", (("c", -1), ("e", -2)), unsigned_out=True))
    open(dst, "w").write(out)
    print("wrote", dst)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("// GENERATED by tools/gen_mul30.py -- do not edit.\n" + gen_fr29() + gen_shoup29() + gen_shoup29_dual())
        print("wrote", sys.argv[2])


if __name__ == "__main__":
    main(sys.argv[1])
