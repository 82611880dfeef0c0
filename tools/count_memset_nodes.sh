#!/bin/bash
# memset (blit) launches per replayed training step: see tools/count_memset_nodes.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/memset_nodes; rm -rf $O; mkdir -p $O
for m in prior vqvae; do
  for n in 4 24; do
    rocprofv3 --kernel-trace -d $O/$m$n -o f -- python3 tools/count_memset_nodes.py $m $n > $O/$m$n.log 2>&1
    python tools/prof_summary.py $O/$m$n/f_results.db 400 > $O/$m$n.txt 2>&1
    rm -rf $O/$m$n
  done
  a=$(grep -m1 "fillBufferAligned" $O/${m}4.txt | awk '{print $2}'); b=$(grep -m1 "fillBufferAligned" $O/${m}24.txt | awk '{print $2}')
  echo "$m: __amd_rocclr_fillBufferAligned launches with 4 / 24 replays: ${a:-0} / ${b:-0}  ->  $(( (${b:-0} - ${a:-0}) / 20 )) per replayed step"
done
