#!/bin/bash
# One GPU-box visit (round 6).  Stages are chosen with STAGES="smoke tests bench prof pmc sq variants pieces ..."; everything
# under timeouts, everything written to gpurun_out/r06/.
STAGES=${STAGES:-"smoke tests bench prof"}
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
cd $GRAFT_REPO_ROOT
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has smoke; then
  timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
if has tests; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest ${TEST_PATHS:-tests} -m gpu -x -q --timeout=900 ${TEST_ARGS:-} > $O/pytest_gpu${TEST_TAG:-}.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu${TEST_TAG:-}.log
fi
if has variants; then
  # coefficient kernel / encoder variants built by tools/build_variants.sh: time + checksum per library
  timeout 900 python tools/time_encode_variants.py > $O/variants${VARIANT_TAG:-}.log 2>&1; echo "variants rc=$?"; cat $O/variants${VARIANT_TAG:-}.log
fi
if has pieces; then
  timeout 600 python tools/time_encode_pieces.py --channels 4096 --pieces 0 4 6 8 12 16 > $O/encode_pieces_4096.log 2>&1; echo "pieces rc=$?"; cat $O/encode_pieces_4096.log
fi
if has channels; then
  timeout 900 python tools/time_encode_channels.py > $O/channel_scaling.log 2>&1; echo "channels rc=$?"; grep -v amdgpu $O/channel_scaling.log
fi
if has coefsvariants; then
  timeout 600 python tools/time_coefs_variants.py --channels 4096 1024 256 1 > $O/coefs_variants.log 2>&1; echo "coefs variants rc=$?"; cat $O/coefs_variants.log
fi
if has bench; then
  for c in ${CODECS:-gc adx hca}; do
    timeout 900 python bench.py --codec $c --steps ${BENCH_STEPS:-5} --warmup 2 ${BENCH_ARGS:-} > $O/bench_$c.json.log 2> $O/bench_$c.err; echo "bench $c rc=$?"; tail -c 3000 $O/bench_$c.json.log; tail -3 $O/bench_$c.err
  done
fi
if has default; then
  ( time timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench default rc=$?"; cat $O/bench_default.time; tail -c 2500 $O/bench_default.json.log
fi
if has ranks8; then
  timeout 900 python bench.py --gpus 8 --share-gpu --channels 512 --seconds 10 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8ranks_shared.json.log 2> $O/bench_8ranks_shared.err; echo "8 ranks rc=$?"; tail -c 3000 $O/bench_8ranks_shared.json.log; tail -5 $O/bench_8ranks_shared.err
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${CODECS:-gc adx hca}; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o r06 -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-signals ${PROF_ARGS:---no-mixed} > $O/prof_$c.log 2>&1; echo "rocprof $c rc=$?"
    f=$(find $O/prof_$c -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-220 && cp $f $O/kernel_stats_$c.csv
  done
  cd $GRAFT_REPO_ROOT
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${PMC_CODECS:-gc adx hca}; do
    for k in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $O/pmc_${c}_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-other-configs --no-mixed --no-signals > $O/pmc_${c}_$k.log 2>&1; echo "pmc $c $k rc=$?"
    done
  done
  cd $GRAFT_REPO_ROOT
  python tools/summarize_pmc.py traffic $O/r06_pmc_traffic.json $(for c in ${PMC_CODECS:-gc adx hca}; do echo $O/pmc_${c}_FETCH_SIZE $O/pmc_${c}_WRITE_SIZE; done) | tail -60
  find $O -name "pmc_*" -type d -exec rm -rf {} + 2>/dev/null
fi
if has sq; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${SQ_CODECS:-gc adx hca}; do
    timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq_${c}_a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-configs --no-mixed --no-signals > $O/sq_${c}_a.log 2>&1; echo "sq $c a rc=$?"
    timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 --kernel-trace --output-format csv -d $O/sq_${c}_b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-configs --no-mixed --no-signals > $O/sq_${c}_b.log 2>&1; echo "sq $c b rc=$?"
  done
  cd $GRAFT_REPO_ROOT
  python tools/summarize_pmc.py sq $O/r06_sq_counters.json $(for c in ${SQ_CODECS:-gc adx hca}; do echo $O/sq_${c}_a $O/sq_${c}_b; done) | tail -80
  find $O -name "sq_*" -type d -exec rm -rf {} + 2>/dev/null
fi
ls $O | head -60
