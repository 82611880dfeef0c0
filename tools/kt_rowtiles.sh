#!/bin/bash
# row-tile launches of batched decoding (B = 32) by grid, for the default library and the ablation builds given in LIBS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
L=$PWD/interactive-spectrogram-inpainting_amd
for lib in ${LIBS:-lib}; do
  O=gpurun_out/kt_rt_$lib; rm -rf $O; mkdir -p $O
  ISI_HIP_LIBRARY=$L/$lib/libisi_hip.so rocprofv3 --kernel-trace -d $O/kt -o f -- python3 tools/prof_sampling.py 32 > $O/log.txt 2>&1
  echo "== $lib"; python tools/prof_by_grid.py $O/kt/f_results.db row_mfma decode_f32 combine | head -14
  rm -rf $O
done
