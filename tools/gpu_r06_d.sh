#!/bin/bash
# round 6, call d: every GPU test that touches HCA on the wave-per-frame encoder, then SQ counters of the HCA bench line
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in test_gpu_hca test_gpu_ragged test_gpu_containers test_gpu_crypt test_gpu_host_pipeline; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu > $O/r06_d_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_d_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_d_pytest_$t.log | cut -c1-250 | head -20
done
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -m gpu -k "hca" > $O/r06_d_pytest_full_hca.log 2>&1
echo "== full size hca: $(grep -v amdgpu.ids $O/r06_d_pytest_full_hca.log | tail -1)"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_hca/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_a.log 2>&1; echo "a rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_hca/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_b.log 2>&1; echo "b rc=$?"
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py sq $O/r06_d_sq_counters_hca.json $O/pmc_sq_hca/a $O/pmc_sq_hca/b | grep -A22 hca_encode_wave
