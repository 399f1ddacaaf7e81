#!/bin/bash
# round 5: the MFMA question once more on this round's code (VERDICT r04 housekeeping), the streaming HCA test after its last edit
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 tools/variants/bench_hca_mfma 2097152 5 > $O/hca_mfma.json.log 2> $O/hca_mfma.err; echo "mfma rc=$?"; cat $O/hca_mfma.json.log | cut -c1-800
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq_mfma -o pmc -- $GRAFT_REPO_ROOT/tools/variants/bench_hca_mfma 2097152 2 > $O/sq_mfma.log 2>&1; echo "mfma pmc rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
out = {}
for f in glob.glob("gpurun_out/r05/sq_mfma/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "mfma" if "mfma" in r["Kernel_Name"] else ("staged" if "staged" in r["Kernel_Name"] else None)
        if k:
            d = out.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_rows"] = d.get("_rows", 0) + 1
json.dump(out, open("gpurun_out/r05/hca_mfma_counters.json", "w"), indent=1)
print(json.dumps(out)[:900])
PY
rm -rf $O/sq_mfma
timeout 600 python -m pytest tests/test_gpu_hca.py -q -m gpu 2>&1 | grep -v amdgpu | tail -2
