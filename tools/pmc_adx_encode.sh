#!/bin/bash
# Counters of the ADX encoder's kernels at configs[2] (tools/adx_encode_once.py): one SQ pass, one pass per TCC counter.
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_adx; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -o pmc -- python $GRAFT_REPO_ROOT/tools/adx_encode_once.py > $O/sq.log 2>&1; echo "sq rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o pmc -- python $GRAFT_REPO_ROOT/tools/adx_encode_once.py > $O/$c.log 2>&1; echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0, 'tools')
import summarize_pmc as S
sq, fs, ws = S.read('gpurun_out/pmc_adx/sq'), S.read('gpurun_out/pmc_adx/FETCH_SIZE'), S.read('gpurun_out/pmc_adx/WRITE_SIZE')
for k, c in sq.items():
    if 'adx_encode' not in k: continue
    ms = c['_dur_ms']
    clock = c['SQ_BUSY_CYCLES'] / 32 / (ms * 1e-3)              # 32 SQ instances count busy cycles (LABNOTES 8.2)
    print("%s: %.2f ms  clock %.2f GHz  waves %d  VALU issue %.3f of the launch, %.3f of the waves' lifetime  wave-instr %.2f G  waiting %.2f"
          "  fetched %.2f GB  written %.2f GB" % (
        k, ms, clock / 1e9, c['SQ_WAVES'], c['SQ_ACTIVE_INST_VALU'] / (1024 * clock * ms * 1e-3 / 4),
        c['SQ_ACTIVE_INST_VALU'] * 4 / c['SQ_WAVE_CYCLES'], c['SQ_INSTS_VALU'] / 1e9, c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'],
        fs.get(k, {}).get('FETCH_SIZE', 0) * 1024 * 2 / 1e9, ws.get(k, {}).get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
