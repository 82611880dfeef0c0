cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt_fwd -o fwd -- python3 bench.py --no-cpu-baseline --no-prior --no-train > $O/bench_under_rocprof.json 2>/dev/null
python tools/prof_summary.py $O/kt_fwd/fwd_results.db 0 > $O/fwd_summary.txt 2>&1
rocprofv3 --kernel-trace -d $O/kt_vt -o vt -- python3 tools/bench_train.py > $O/vt.log 2>&1
python tools/prof_summary.py $O/kt_vt/vt_results.db 0 > $O/vqvae_train_summary.txt 2>&1
python tools/bench_train.py > $O/vqvae_train.txt 2>&1
rm -rf $O/kt_*
tail -c 200 $O/bench.json
