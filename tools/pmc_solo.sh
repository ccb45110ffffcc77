#!/bin/bash
# PMC passes over the solo run (one lane, every launch alone on the GPU, all images live for the first
# iterations): per-kernel counter averages, printed and kept as gpurun_out/pmc_solo_summary.txt.
# Usage: tools/pmc_solo.sh "<solo_profile.py args>"
R=$GRAFT_REPO_ROOT
ARGS=${1:-"4096 60"}
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcs_$name -- python3 $R/tools/solo_profile.py $ARGS > /tmp/pmcs_$name.log 2>&1
  echo "pass $name exit=$?"
}
run valu SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY || exit 1
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
python3 - <<PY | tee $R/gpurun_out/pmc_solo_summary.txt
import csv, glob, collections
print("solo_profile.py $ARGS: every launch alone on the GPU; per-launch averages of each counter")
for name in ("valu", "lds", "grbm"):
    files = glob.glob("/tmp/pmcs_%s/*/*counter_collection.csv" % name)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in agg:
        if any(s in k for s in ("k_search", "k_priority", "k_update", "k_select")):
            n = len(disp[k])
            print(name, k, "launches", n, {c: round(v / n, 1) for c, v in agg[k].items()})
PY
