cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gcadpcm.py tests/test_gpu_golden.py tests/test_gpu_full_size.py -m gpu -x -q --timeout=600 > $O/pytest_gc.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gc.log
timeout 600 python tools/time_encode_pieces.py --channels 128 256 384 512 1024 --pieces 0 32 64 128 256 512 > $O/encode_pieces.log 2>&1; cat $O/encode_pieces.log
