"""roofline.traffic / paths.ntt_roofline.traffic: HBM bytes per launch from the PMC counters, collected live -- during the bench run --
by child rocprofv3 passes, the way MI355X_MICROARCH.md prescribes: one pass per counter (--kernel-trace --pmc FETCH_SIZE, then
WRITE_SIZE: they do not fit one pass on gfx950), FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide loads at 64 bytes),
units of KB.  The profiled program is `python3 bench.py --pmc-child <log_n> --pmc-kind msm|ntt` (python3 itself behind `--`: no shell
hop, no exec after the GPU is initialised)."""
import ctypes
import os
import sys
import tempfile

from .common import BENCH_PY, SEED, TAU

KERNELS = {"msm": ("k_accum_affine",), "ntt": ("k_ntt_tile", "k_ntt_pass")}


def pmc_child(log_n, kind="msm"):
    """The workload of one rocprofv3 pass.  msm: a few lone degree-2^log_n commitments on uniform scalars resident in HBM --
    k_accum_affine launches of exactly the shape the timed region runs.  ntt: forward transforms of 2^log_n resident scalars."""
    import kzg_amd
    from kzg_amd import _lib as L
    e = kzg_amd.Engine(0)
    n = 1 << log_n
    sc = e.alloc_scalars(n).fill_random(SEED)
    if kind == "ntt":
        for _ in range(6):
            assert e.lib.kzg_ntt_fr(e.ctx, sc.ptr, log_n, 0, L.IN_DEVICE) == 0, e.last_error()
    else:
        params = kzg_amd.setup(e, TAU, n, g2_len=0)
        out = ctypes.create_string_buffer(96)
        for _ in range(4):
            assert e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, sc.ptr, n, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, e.last_error()
        params.gs.free()
    sc.free()
    e.close()


def measure_traffic_pmc(log_n, kind="msm", timeout_s=150):
    """HBM bytes per launch (msm: of k_accum_affine; ntt: per kernel and per transform = the sum over the transform's kernels).
    Returns (dict or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp:
        return None, "rocprofv3 not found"
    want = KERNELS[kind]
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="kzg_pmc_", dir="/tmp")
        try:
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, BENCH_PY,
                   "--pmc-child", str(log_n), "--pmc-kind", kind]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if any(w in name for w in want) and row.get("Counter_Name") == ctr:
                        key = name.split("(")[0].replace("void ", "").replace("kzg::", "").strip()
                        a = per.setdefault(key, [0.0, 0])
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
            if not per:
                return None, "%s pass produced no %s rows (rc %d): %s" % (ctr, "/".join(want), r.returncode, (r.stderr or "")[-200:])
            vals[ctr] = {k: (v[0] / v[1], v[1]) for k, v in per.items()}
        except Exception as e:  # noqa: BLE001
            return None, "%s pass failed: %s" % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    method = ("two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of a child process running %s, collected during this "
              "bench run; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction of MI355X_MICROARCH.md)"
              % (("lone 2^%d commitments" if kind == "msm" else "forward 2^%d transforms") % log_n))
    if kind == "msm":
        (f_kb, cnt), (w_kb, _) = list(vals["FETCH_SIZE"].values())[0], list(vals["WRITE_SIZE"].values())[0]
        return {"bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)), "raw_fetch_kb": round(f_kb, 1), "raw_write_kb": round(w_kb, 1),
                "launches_sampled": cnt, "method": method}, None
    kernels = {}
    for k in sorted(set(vals["FETCH_SIZE"]) & set(vals["WRITE_SIZE"])):
        f_kb, w_kb = vals["FETCH_SIZE"][k][0], vals["WRITE_SIZE"][k][0]
        kernels[k] = {"bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)), "raw_fetch_kb": round(f_kb, 1), "raw_write_kb": round(w_kb, 1),
                      "launches_sampled": vals["FETCH_SIZE"][k][1]}
    return {"bytes_per_transform": sum(v["bytes_per_launch"] for v in kernels.values()), "kernels": kernels, "method": method}, None
