cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout 600 python tools/time_encode_pieces.py --channels 1 8 32 64 96 --pieces 0 16 32 64 128 256 > $O/encode_pieces_small.log 2>&1; cat $O/encode_pieces_small.log
