#!/bin/bash
# round 6, call b: stage times and occupancy variants of the wave-per-frame HCA encoder, SQ counters of the HCA bench
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 600 python tools/time_hca_decode.py > $O/r06_b_hca_encode_stages.log 2>&1
grep -v amdgpu.ids $O/r06_b_hca_encode_stages.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_hca/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_a.log 2>&1; echo "a rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_hca/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_b.log 2>&1; echo "b rc=$?"
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py sq $O/r06_b_sq_counters_hca.json $O/pmc_sq_hca/a $O/pmc_sq_hca/b | head -60
