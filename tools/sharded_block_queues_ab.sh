#!/bin/bash
# round 6: the device group at world 1 (RCCL all-gather forced on), one step = kzg_commit_coeff_sharded_batch of 64 polynomials of 2^20 coefficients:
# the shipped library against the build before the group had an exchange stream of its own (tools/bin/lib_preexch.so), and the shipped one with 14 lanes
# (the round-5 plan: the exchange stream is then the 25th stream of a 24-queue process)
for rep in 1 2; do
  python tools/group_step_probe.py 2>&1 | grep "ms per step"
  PROBE_STREAMS=14 python tools/group_step_probe.py 2>&1 | grep "ms per step" | sed 's/library shipped/library shipped, streams=14/'
  KZG_AMD_LIBRARY=$PWD/tools/bin/lib_preexch.so python tools/group_step_probe.py 2>&1 | grep "ms per step"
done
