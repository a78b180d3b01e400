import torch, os, sys, runpy
sys.argv=["bench.py","--no-cpu-baseline","--no-paths","--steps","4","--sharded-block","--sharded-steps","4"]
sys.path.insert(0, os.getcwd())
import kzg_amd; kzg_amd.load()
if os.environ.get("WITH_PG"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29561")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda",0), rank=0, world_size=1)
    dist.barrier()
runpy.run_path("bench.py", run_name="__main__")
