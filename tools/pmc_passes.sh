#!/bin/bash
# Separate rocprofv3 PMC passes over a short bench run (one counter group per run, no tracing
# domains beside it).  Outputs under gpurun_out/pmc_<name>/.  Usage: tools/pmc_passes.sh [kernel_stats.csv of
# the same command: its average launch durations go into the stencil-stage summary]
R=$GRAFT_REPO_ROOT
KSTATS=${1:-}
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --configs "" --no-host-io --no-split > $R/gpurun_out/pmc_$name.log 2>&1
  echo "pass $name exit=$?"
}
run valu SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY || exit 1
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
# (a GRBM pass over the whole bench ran into its time limit on some boxes and is left out; tools/pmc_solo.sh has one)
python3 - <<PY
import csv, glob, collections, json
summary = {}
for name in ("valu", "lds", "fetch", "write"):
    files = glob.glob("$R/gpurun_out/pmc_%s/*/*counter_collection.csv" % name)
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
            summary.setdefault(k, {})["launches_" + name] = n
            for c, v in agg[k].items():
                summary[k][c + "_per_launch"] = v / n
            print(name, k, "launches", n, {c: round(v / n, 1) for c, v in agg[k].items()})
json.dump(summary, open("$R/gpurun_out/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in summary.items():
    if "k_search" in k and "FETCH_SIZE_per_launch" in v and "WRITE_SIZE_per_launch" in v:
        json.dump({"kernel": k, "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE_per_launch"],
                   "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE_per_launch"],
                   "launches": v.get("launches_fetch"),
                   "images_per_launch": 4096,   # default bench batch 8192 on 2 lanes
                   "command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --configs \"\" --no-host-io --no-split"},
                  open("$R/gpurun_out/pmc_search_latest.json", "w"), indent=1)
# the stencil stages (priority, update) with their launch durations from the kernel trace of the same command
dur, share = {}, {}
if "$KSTATS":
    try:
        for r in csv.DictReader(open("$KSTATS")):
            k = r["Name"].split("(")[0]
            dur[k] = float(r["AverageNs"]) / 1e3
            share[k] = float(r["Percentage"]) / 100.0
    except OSError:
        pass
sten = {}
for k, v in summary.items():
    if ("k_priority" in k or "k_update" in k or "k_step" in k) and "FETCH_SIZE_per_launch" in v and "WRITE_SIZE_per_launch" in v and k in dur:
        sten[k.replace("void ", "")] = {"FETCH_SIZE_KB_per_launch": v["FETCH_SIZE_per_launch"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE_per_launch"],
                                        "avg_launch_us": dur[k], "share_of_gpu_time": share.get(k), "images_per_launch": 4096,
                                        "salu_per_valu": (v["SQ_INSTS_SALU_per_launch"] / v["SQ_INSTS_VALU_per_launch"]) if v.get("SQ_INSTS_VALU_per_launch") else None}
if sten:
    json.dump({"kernels": sten, "command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* (separate runs) and --kernel-trace --stats -- python3 bench.py --configs \"\" --no-host-io --no-split --no-cpu-baseline (headline batch, 4096 images per launch)"},
              open("$R/gpurun_out/pmc_stencil_latest.json", "w"), indent=1)
PY
