"""Prints name, calls, total ms and average us of the iteration kernels from rocprofv3 kernel_stats.csv files."""
import csv, sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_priority", "k_search", "k_update", "k_step", "k_select", "k_local_best", "k_nf_", "k_tiles", "k_init_")):
            print("%-46s calls %6s total %9.2f ms avg %9.2f us" % (r["Name"].replace("void ", "").replace("(Params)", ""),
                  r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
