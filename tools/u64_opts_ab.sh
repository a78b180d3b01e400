#!/bin/bash
# round 6: u64-valued coefficients at 2^20, pipeline options against the default (tools/u64_probe.py prints commitments/s and the in-situ kernel times)
O=gpurun_out/r06_u64; mkdir -p $O; rm -f $O/opts.txt
for rep in 1 2; do
for opts in "" "streams=20" "streams=24" "accum_streams=3" "accum_streams=4" "small_entries=4194304" "small_entries=4194304 accum_blocks_small=240" "small_entries=4194304 accum_blocks_small=320 accum_streams_small=3" "accum_blocks_batch=512" "accum_blocks_batch=448" "defer_tail=0" "sort_threads_batch=512"; do
  python tools/u64_probe.py $opts 2>&1 | grep -E "^options|k_accum_affine" >> $O/opts.txt
done
done
cat $O/opts.txt
