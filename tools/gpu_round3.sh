#!/bin/bash
# One GPU-box visit (round 3).  Stages are chosen with STAGES="smoke tests bench prof pmc sq"; everything under timeouts,
# everything written to gpurun_out/r03/.
STAGES=${STAGES:-"smoke tests bench prof"}
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
cd $GRAFT_REPO_ROOT
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has hcatests; then
  timeout 1200 python -m pytest tests/test_gpu_hca.py tests/test_gpu_golden.py tests/test_gpu_full_size.py -m gpu -x -q --timeout=600 -k "hca or Hca" > $O/pytest_hca.log 2>&1; echo "pytest hca rc=$?"; tail -15 $O/pytest_hca.log
fi
if has smoke; then
  timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
if has tests; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --timeout=600 ${TEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
fi
if has bench; then
  for c in ${CODECS:-gc adx hca}; do
    timeout 900 python bench.py --codec $c --steps ${BENCH_STEPS:-5} --warmup 2 > $O/bench_$c.json.log 2> $O/bench_$c.err; echo "bench $c rc=$?"; tail -c 1500 $O/bench_$c.json.log
  done
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${CODECS:-gc adx hca}; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o r03 -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/prof_$c.log 2>&1; echo "rocprof $c rc=$?"
    f=$(find $O/prof_$c -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-220
  done
  cd $GRAFT_REPO_ROOT
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${PMC_CODECS:-gc adx hca}; do
    for k in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $O/pmc_${c}_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 2 --warmup 0 --no-cpu-baseline --no-e2e > $O/pmc_${c}_$k.log 2>&1; echo "pmc $c $k rc=$?"
    done
  done
  cd $GRAFT_REPO_ROOT
  python tools/summarize_pmc.py traffic $O/r03_pmc_traffic.json $O/pmc_gc_FETCH_SIZE $O/pmc_gc_WRITE_SIZE $O/pmc_adx_FETCH_SIZE $O/pmc_adx_WRITE_SIZE $O/pmc_hca_FETCH_SIZE $O/pmc_hca_WRITE_SIZE | tail -60
  # the raw per-dispatch CSVs are tens of MB: keep the summaries only
  find $O -name "pmc_*" -type d -exec rm -rf {} + 2>/dev/null
fi
if has sq; then
  cd /tmp && export TMPDIR=/tmp
  for c in ${SQ_CODECS:-gc adx hca}; do
    timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq_${c}_a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $O/sq_${c}_a.log 2>&1; echo "sq $c a rc=$?"
    timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 --kernel-trace --output-format csv -d $O/sq_${c}_b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec $c --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $O/sq_${c}_b.log 2>&1; echo "sq $c b rc=$?"
  done
  cd $GRAFT_REPO_ROOT
  python tools/summarize_pmc.py sq $O/r03_sq_counters.json $O/sq_gc_a $O/sq_gc_b $O/sq_adx_a $O/sq_adx_b $O/sq_hca_a $O/sq_hca_b | tail -80
  find $O -name "sq_*" -type d -exec rm -rf {} + 2>/dev/null
fi
if has mfma; then
  timeout 300 tools/variants/bench_hca_mfma 2097152 5 > $O/hca_mfma.json.log 2> $O/hca_mfma.err; echo "mfma rc=$?"; cat $O/hca_mfma.json.log
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq_mfma -o pmc -- $GRAFT_REPO_ROOT/tools/variants/bench_hca_mfma 2097152 2 > $O/sq_mfma.log 2>&1; echo "mfma pmc rc=$?"
  cd $GRAFT_REPO_ROOT
  python - <<'PY'
import csv, glob, json, collections
rows = []
for f in glob.glob("gpurun_out/r03/sq_mfma/**/*counter_collection*.csv", recursive=True):
    rows += list(csv.DictReader(open(f, newline="")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = "mfma" if "mfma" in r["Kernel_Name"] else ("staged" if "staged" in r["Kernel_Name"] else None)
    if k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/r03/hca_mfma_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  rm -rf $O/sq_mfma
fi
ls $O | head -40
