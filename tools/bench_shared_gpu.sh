#!/bin/bash
# The driver's N > 1 launch of bench.py with all N ranks on the ONE GPU of a test box: the bench's TCP-star control plane (no torch in the ranks), and the device group over
# the hooks build's test transport (kzg_amd/csrc/test_transport.h; RCCL refuses two ranks on one GPU).  Numbers from this are not
# scaling numbers -- the ranks share a GPU -- it shows that the N > 1 line is produced and checked.
#   bash tools/bench_shared_gpu.sh N [bench.py arguments...]
N=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export KZG_BENCH_SHARED_GPU=1 KZG_TEST_SHM_TRANSPORT=1 KZG_AMD_LIBRARY=$PWD/kzg_amd/libkzg_mi355x_hooks.so MASTER_ADDR=127.0.0.1
exec python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus $N "$@"
