"""A/B of the one-node RCCL environment knobs on the time to form a world-1 device group and run its first exchange, each
configuration in a fresh child process (RCCL reads its environment once).  Output: one line per configuration with the library's
own phase timings (kzg_mctx_info).  Usage: python tools/rccl_formation_ab.py [reps]  ->  profiles/r05_rccl_formation_ab.txt"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
t0 = time.time()
sys.path.insert(0, %r)
import kzg_amd
g = kzg_amd.DeviceGroup([0])
g.set_option("always_gather", 1)
s = g.setup(5, 1024)
c = g.commit(s, list(range(1, 1025)))
t1 = time.time()
c2 = g.commit(s, list(range(1, 1025)))
t2 = time.time()
f = g.formation()
s.free()
g.close()
print(json.dumps({"total_s": round(time.time() - t0, 3), "to_first_commit_s": round(t1 - t0, 3), "second_commit_ms": round((t2 - t1) * 1e3, 2),
                  "phases_ms": f, "same": c == c2}))
''' % ROOT

KNOBS = {"NCCL_SOCKET_IFNAME": "lo", "NCCL_RAS_ENABLE": "0", "NCCL_IB_DISABLE": "1", "NCCL_NET_PLUGIN": "none",
         "RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0"}


def run(name, extra, reps):
    for i in range(reps):
        env = {k: v for k, v in os.environ.items() if k not in KNOBS}
        env["KZG_RCCL_SINGLE_NODE_ENV"] = "0"
        env.update(extra)
        t = time.time()
        try:
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=420)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res = json.loads(line[-1]) if line else {"rc": r.returncode, "stderr": r.stderr[-400:]}
        except subprocess.TimeoutExpired:
            res = {"timeout_s": 420}
        print(json.dumps({"config": name, "rep": i, "wall_s": round(time.time() - t, 2), **res}), flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    run("none (RCCL defaults)", {}, reps)
    for k, v in KNOBS.items():
        run("%s=%s only" % (k, v), {k: v}, reps)
    run("host default set (kzg_amd.distributed.SINGLE_NODE_RCCL_ENV)", {"KZG_RCCL_SINGLE_NODE_ENV": "1"}, reps)
    run("all six", dict(KNOBS), reps)


if __name__ == "__main__":
    main()
