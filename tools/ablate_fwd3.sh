cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in "" a1 a2 a4 a7 a8 a16 a32 a63; do
  lib=interactive-spectrogram-inpainting_amd/lib${v:+_$v}/libisi_hip.so
  O=gpurun_out/kt_abl; rm -rf $O; mkdir -p $O
  ISI_HIP_LIBRARY=$PWD/$lib rocprofv3 --kernel-trace --stats -d $O/kt -o a -- python3 tools/check_attention_fwd3.py --no-sweep --modes 1 --precs bf16x3 > $O/log.txt 2>&1
  echo "variant ${v:-base}: $(python tools/prof_summary.py $O/kt/a_results.db 6 2>/dev/null | grep fwd3_kernel | awk '{print $(NF-1)}') us"
done
rm -rf gpurun_out/kt_abl
