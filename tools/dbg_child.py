import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode in ("hip", "hip_batch"):
    import kzg_amd, ctypes
    from kzg_amd import _lib as L
    e = kzg_amd.Engine(0)
    if mode == "hip_batch":
        n = 1 << 20
        p = kzg_amd.setup(e, 5, n, g2_len=0); sc = e.alloc_scalars(n * 16).fill_random(1); out = ctypes.create_string_buffer(96 * 16)
        assert e.lib.kzg_msm_g1_batch(e.ctx, p.gs.handle, 0, sc.ptr, n, 16, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
        sc.free(); p.gs.free()
    e.close()
env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
extra = {}
if len(sys.argv) > 2:
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1); extra[k] = v
env.update(extra)
kw = {}
if os.environ.get("NEWSESSION"): kw["start_new_session"] = True
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-child", "--gpus", "1", "--log-n", "20", "--sharded-batch", "64", "--sharded-steps", "3", "--streams", "0"],
                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)
d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
print(mode, extra, {k: d[k]["value"] for k in ("strong", "config5") if k in d}, d.get("formation", {}).get("init"))
