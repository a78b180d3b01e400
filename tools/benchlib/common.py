"""Constants and helpers shared by the bench modules.  No oracle, no GPU call at import."""
import ctypes
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH_PY = os.path.join(ROOT, "bench.py")

LOG_N = 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_TERM = 128           # 32 B scalar + 96 B affine point per MSM term
NTT_BYTES_PER_ELEM = 64        # one read + one write of a 32-byte Fr
# The resource that binds k_accum_affine: VALU issue, dominated by v_mad_i64_i32.  One bucket addition
# executes 6 mul30 (338 mads) + 2 sqr30 (260) + one fused double product with a single reduction (507) = 3055 mads per lane
# (static count from the ISA).  tools/microbench.hip measured the chip's 64-bit multiply-add issue rate: 31.5 T lane-op/s at
# 8 waves/SIMD, 23.7 T at the 2 waves/SIMD a 200-VGPR kernel holds (profiles/r01_microbench.txt).
MADS_PER_ADD = 6 * 338 + 2 * 260 + 507
MAD_PEAK_TLANE_S = 31.51       # reference value (round-1 microbench on another box); the line's `peak` is measured in this run
MAD_PEAK_OCC2_TLANE_S = 23.70
MAD_NOMINAL_TLANE_S = 256 * 4 * 64 / 4 * 2.4e9 / 1e12   # 39.3: 256 CUs x 4 SIMDs x 64 lanes, one wave-instruction per 4 cycles, 2.4 GHz nominal
FR_MUL_PEAK_G_S = 111.0        # measured Fr (9 x 29-bit) multiplies per second of the NTT's multiply
MADS_PER_FR29_MUL = 162        # 9 x 9 products + 9 x 9 reduction products of one Fr29 Montgomery multiply (fr29.h): the NOMINAL price of a
                               # butterfly multiplication, kept so that mad_frac stays comparable with earlier rounds
MADS_PER_SHOUP_MUL = 143       # what the stage twiddles cost since round 4: 53 (quotient columns) + 45 + 45 multiply-adds (fr29.h)
TAU = 0x5EED5EED5EED5EED       # known secret for the synthetic SRS (setup(s, n), src/lib.rs:38)
SEED = 1


def g1_adds_per_msm(n, c, W):
    """algorithmic G1 additions, n*W bucket accumulations + bucket reduction (W: digits per scalar; positional
    tables, c = 18: the measured average number of NAF digits, 2^16 buckets)."""
    return n * W + 2 * (1 << ((17 if c == 18 else c) - 1))


def naf18_avg_digits(blob):
    """Average number of width-18 NAF digits (kzg_amd/csrc/naf.h) of the canonical 32-byte scalars in `blob`: what one scalar
    contributes to the sorted entry list when the SRS uses positional tables."""
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    tot, cnt = 0, 0
    for o in range(0, len(blob), 32):
        k = int.from_bytes(blob[o:o + 32], "little") % R
        if k >> 254:
            k = R - k
        while k:
            if k & 1:
                d = k & 0x3ffff
                k -= d - (1 << 18) if d >= (1 << 17) else d
                tot += 1
            k >>= 1
        cnt += 1
    return tot / max(cnt, 1)


def view(kzg_amd, buf, first, n):
    v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    v.engine, v.n, v.sfmt, v.ptr = buf.engine, n, buf.sfmt, ctypes.c_void_p(buf.ptr.value + 32 * first)
    return v


def timeit(f, reps=10, warm=4, warm_ms=40.0):
    """ms per call: the median of `reps` blocking calls, each timed by itself.  Untimed calls first -- at least `warm` of them and at
    least `warm_ms` of them: every reading of `paths` follows seconds of host-side checking (the oracle's Horner loops) during which
    the GPU idles and drops its clocks -- one warm-up call read a lone create_witness 0.3-0.5 ms slower inside the bench than the
    same call in a loop, four still 0.2-0.3 ms slower (profiles/r06_prof_witness_coeff.txt: commit 2.41 but create_witness 2.74 ms in
    one line, 0.04-0.10 ms apart in a loop)."""
    t0, i = time.perf_counter(), 0
    while i < warm or ((time.perf_counter() - t0) * 1e3 < warm_ms and i < 200):
        f()
        i += 1
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return (ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])) * 1e3


def view_of(buf, first, n):
    import kzg_amd
    return view(kzg_amd, buf, first, n)


