cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round; mkdir -p $O
SECONDS=0
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench.py wall: ${SECONDS}s" > $O/bench_wall.txt
rocprofv3 --kernel-trace --stats -d $O/kt_fwd -o fwd -- python3 bench.py --no-cpu-baseline --no-prior --no-train > $O/bench_under_rocprof.json 2>/dev/null
python tools/prof_summary.py $O/kt_fwd/fwd_results.db 0 > $O/fwd_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_vt -o vt -- python3 tools/bench_train.py > $O/vt.log 2>&1
python tools/prof_summary.py $O/kt_vt/vt_results.db 0 > $O/vqvae_train_summary.txt 2>&1
python tools/bench_train.py > $O/vqvae_train.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_ps -o ps -- python3 tools/prof_sampling.py 1 > $O/ps.log 2>&1
python tools/prof_summary.py $O/kt_ps/ps_results.db 0 > $O/prior_sampling_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_ps32 -o ps -- python3 tools/prof_sampling.py 32 > $O/ps32.log 2>&1
python tools/prof_summary.py $O/kt_ps32/ps_results.db 0 > $O/prior_sampling_b32_summary.txt 2>&1
rm -rf $O/kt_*
tail -c 200 $O/bench.json
