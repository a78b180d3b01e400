"""The `sharded` block: north_star's sharded-SRS + RCCL design (device group), measured by the ranks' fresh child processes."""
import ctypes
import json
import os
import sys
import tempfile
import time

from . import checks
from .common import *  # noqa: F401,F403
from .control import Job


def measure_sharded_block(kzg_amd, L, job, args, force_gather):
    """What north_star names, measured in the default multi-GPU run next to the replicas: the SRS sharded over the ranks, one
    partial point per rank and polynomial, ONE RCCL all-gather of the 144-byte partials inside the library, local sums
    (kzg_commit_coeff_sharded_batch, kzg_amd/csrc/mgpu.hip).  (i) strong: every degree-2^log_n commitment sharded N ways;
    (ii) config5: BASELINE configs[4], 2^21 terms per rank (degree 2^24 at N = 8).  Every commitment of the last step of each is
    checked against [p(tau)]G by the oracle.  A group that cannot form degrades to a note."""
    from kzg_amd.api import DeviceGroup
    from kzg_amd.distributed import shard_range
    rank, world = job.rank, job.world
    res = {}
    group, err = None, None
    try:
        uid = job.broadcast_object(DeviceGroup.unique_id)
        group = DeviceGroup.for_rank(job.local_rank, rank, world, uid)
        if force_gather:
            group.set_option("always_gather", 1)
    except Exception as e:  # noqa: BLE001
        err = str(e)
    if not job.all_agree(group is not None):
        if group is not None:
            group.close()
        return {"note": "device group could not be formed (%s): sharded-SRS + RCCL modes not measured in this run" % (err or "failure on another rank")}
    try:
        res["rccl"] = group.info()
        res["rccl_ranks"] = group.world
        eng = group.engine(0)
        if args.streams:
            eng.set_option("streams", args.streams)
        job.engines.append(eng)
        batch, steps = args.sharded_batch, args.sharded_steps
        for mode, n_poly in (("strong", 1 << args.log_n), ("config5", world << 21)):
            lo, hi = shard_range(n_poly, rank, world)
            n_local = hi - lo
            scal = eng.alloc_scalars(max(n_local, 1) * batch)
            for b in range(batch):
                view(kzg_amd, scal, b * n_local, n_local).fill_random(SEED + 5000 + 1000 * b + 4 * lo)
            msrs = group.setup(TAU, n_poly)
            srs, first = msrs.shard(0)
            assert first == lo and len(srs) == n_local
            c, W = srs.window_info()
            out = ctypes.create_string_buffer(96 * batch)
            ptrs = (ctypes.c_void_p * 1)(scal.ptr.value)

            def step():
                rc = group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n_poly, batch, scal.sfmt, L.IN_DEVICE, out,
                                                              L.G1_AFFINE_MONT)
                if rc:
                    raise RuntimeError(group.last_error())
            step()
            if "formation" not in res:      # after the first exchange: what forming the communicator cost on this rank (ms per phase)
                res["formation"] = group.formation()
                res["rccl"] = group.info()
            job.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            job.barrier()
            dt = job.max_over_ranks(time.perf_counter() - t0)
            ok = checks.check_known_tau(kzg_amd, job, scal, n_local, lo, out.raw, list(range(batch)), spans_ranks=True)
            v = batch * steps / dt
            res[mode] = {"value": round(v, 3), "unit": "commitments/s", "scaling": "strong" if mode == "strong" else "weak",
                         "polynomial_coefficients": n_poly, "terms_per_rank": n_local, "batch": batch, "steps": steps,
                         "ms_per_step": round(dt / steps * 1e3, 4), "window_bits": c, "windows": W,
                         "msm_terms_per_sec": round(v * n_poly, 1), "g1_adds_per_sec": round(v * world * g1_adds_per_msm(n_local, c, W), 1),
                         "hbm_frac_algorithmic": round(BYTES_PER_TERM * n_poly * v / 1e9 / (HBM_PEAK_GBS * world), 6),
                         "collective": "one ncclAllGather of (batch + 1) x 144 B per rank and step, inside the library",
                         "all_results_match_known_tau": ok}
            scal.free()
            msrs.free()
        job.engines.remove(eng)
    except Exception as e:  # noqa: BLE001
        res["error"] = str(e)
    finally:
        group.close()
    return res


def sharded_child_main(args):
    """`bench.py --sharded-child`: the sharded block alone, in a fresh process per rank (started by run_sharded_block_in_children).
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the parent; rank 0 prints the block as one JSON line."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)       # RCCL's banner and anything else native goes to stderr: stdout carries the JSON line only
    job = Job(rank, local_rank, world)
    import kzg_amd
    from kzg_amd import _lib as L
    res = measure_sharded_block(kzg_amd, L, job, args, force_gather=(world == 1))
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    job.barrier()
    job.close()
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    os.close(real_stdout)


def _tail(path_or_text, n=1500, is_path=False):
    try:
        t = open(path_or_text, errors="replace").read() if is_path else (path_or_text or "")
    except OSError:
        return ""
    return t[-n:]


def run_sharded_block_in_children(args, rank, local_rank, world):
    """The `sharded` block in a FRESH child process per rank, run BEFORE this process touches the GPU.  Whatever goes wrong while a
    device group forms over RCCL -- a bootstrap that stalls for minutes on a hostile network stack (round 4's driver box), a crash
    inside the communicator, a dead peer -- happens in a process that can be killed; this process' line and exit status stay
    truthful, and the block says what happened: the library's per-phase formation times (KZG_DEBUG), the child's exit code, and the
    tail of RCCL's own log (NCCL_DEBUG=INFO into NCCL_DEBUG_FILE from the start).  Before, not after: a process that has used the
    GPU slows every OTHER process on it by its mere presence (its hardware queues stay mapped; measured from a parent that had run
    one batch and closed its engine: 43 instead of 460 commitments/s in the child, and hipDeviceReset does not give them back)."""
    import glob
    import signal
    import subprocess
    # the children's own rendezvous: a port every rank can derive without talking (this runs before the ranks have a process group)
    base = int(os.environ.get("MASTER_PORT", "29531"))
    port = base + 29 if base + 29 < 65536 else base - 29
    log_prefix = os.path.join(tempfile.gettempdir(), "kzg_rccl_%d_r%d" % (os.getpid(), rank))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}   # (the agent-store flag would make the child look
    #                                                                                     for torchrun's store on the new port)
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               KZG_DEBUG="1")
    if env.get("KZG_RCCL_SINGLE_NODE_ENV", "1") != "0":     # this process is the host: the one-node RCCL knobs (kzg_amd/distributed.py)
        from kzg_amd.distributed import SINGLE_NODE_RCCL_ENV
        for k, v in SINGLE_NODE_RCCL_ENV.items():
            env.setdefault(k, v)
    if env.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):
        env["NCCL_DEBUG"] = "INFO"
        env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,BOOTSTRAP,NET,ENV")
    env.setdefault("NCCL_DEBUG_FILE", log_prefix + ".%p.log")
    cmd = [sys.executable, BENCH_PY, "--sharded-child", "--gpus", str(world), "--log-n", str(args.log_n),
           "--sharded-batch", str(args.sharded_batch), "--sharded-steps", str(args.sharded_steps), "--streams", str(args.streams)]
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, err = p.communicate(timeout=args.sharded_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)     # exactly the process group this call started
        except OSError:
            pass
        out, err = p.communicate()
    wall = time.perf_counter() - t0
    block = None
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            try:
                block = json.loads(ln)
            except ValueError:
                pass
    child = {"rc": p.returncode, "wall_s": round(wall, 2), "timed_out": timed_out, "process": "fresh child per rank"}
    healthy = not timed_out and p.returncode == 0     # (rank 0's child has agreed every check with the other ranks' children)
    if rank != 0:
        for f in glob.glob(log_prefix + "*"):
            try:
                os.unlink(f)
            except OSError:
                pass
        return None
    if block is None:
        block = {"note": ("the sharded block did not finish within %d s and its process was killed" % args.sharded_timeout) if timed_out
                 else "the sharded block's process ended with code %s and no result" % p.returncode}
    block["child"] = child
    slow = isinstance(block.get("formation"), dict) and block["formation"].get("formation_ms", 0) > 10000
    if not healthy or slow or "error" in block or "note" in block:
        logs = sorted(glob.glob(log_prefix + "*"))
        block["diagnostics"] = {"stderr_tail": _tail(err, 2500), "rccl_log_tail": _tail(logs[0], 2500, is_path=True) if logs else "",
                                "env": {k: v for k, v in env.items() if k.startswith(("NCCL_", "RCCL_", "KZG_", "GPU_MAX"))}}
    for f in glob.glob(log_prefix + "*"):
        try:
            os.unlink(f)
        except OSError:
            pass
    return block


