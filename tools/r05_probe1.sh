set -x
mkdir -p gpurun_out/r05
env | grep -E "NCCL|RCCL|HSA|GPU_|HIP|ROCR|LD_|PYTHON" > gpurun_out/r05/env.txt
hostname; cat /etc/hosts; cat /etc/resolv.conf; ip addr 2>/dev/null | head -30; ls /sys/class/kfd/kfd/topology/nodes | head; ls /sys/class/infiniband 2>&1 | head
(time python tools/rccl_formation_probe.py) > gpurun_out/r05/probe_default.txt 2>&1
(time NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=ALL python tools/rccl_formation_probe.py) > gpurun_out/r05/probe_debug.txt 2>&1
(time python tools/rccl_formation_probe.py /usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so) > gpurun_out/r05/probe_torch_rccl.txt 2>&1
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke_time.txt 2>&1
tail -3 gpurun_out/r05/probe_default.txt gpurun_out/r05/probe_torch_rccl.txt gpurun_out/r05/smoke_time.txt
