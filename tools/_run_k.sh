cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json.log
VGA_HIP_PIPELINE_TIMELINE=1 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-other-configs > $O/bench_timeline.json.log 2> $O/bench_timeline.err; grep timeline $O/bench_timeline.err | tail -12
