#!/bin/bash
# kernel-level timing of a 512-channel GC-ADPCM encode (time pieces + seams) under rocprofv3
NCH=${NCH:-512}; export NCH
mkdir -p gpurun_out/prof_small_$NCH
cat > /tmp/small.py <<'PY'
import sys, torch, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vgaudio_amd import device as vdev
d = torch.device('cuda:0'); nch, n = int(os.environ.get("NCH", "512")), 2880000
pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); out = vdev.alloc_adpcm(nch, n, d)
for _ in range(3):
    vdev.gc_encode(pcm, n, coefs, out=out)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small_$NCH -o s -- python /tmp/small.py > $GRAFT_REPO_ROOT/gpurun_out/prof_small_$NCH/log.txt 2>&1
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/prof_small_$NCH -name "*kernel_stats*.csv" | head -1); do cut -c1-200 $f | head -8; done
