mkdir -p gpurun_out/r02
L=gpurun_out/r02/sweep_pipeline16.log; : > $L
run() { echo "## $1 $2" >> $L; env $1 timeout 200 python tools/sweep_host_pipeline.py --shapes "$2" 2>&1 | grep -v "amdgpu.ids" | cut -c1-1000 | tail -9 >> $L; }
run "VGA_HIP_PIPELINE_TIMELINE=1" "0,0,0,0"
run "X=1" "0,0,0,-1;1,3,0,0"
cat $L
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_gcadpcm.py tests/test_gpu_adx.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-3000
