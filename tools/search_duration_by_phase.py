"""Splits the search kernel's launches in a rocprofv3 kernel trace of `python3 bench.py` into the
phases of the bench (graph replays of warm-up + timed steps, the exclusive-GPU event pass, the
overlapped event pass) and prints the average duration of each.
Usage: python tools/search_duration_by_phase.py <kernel_trace.csv> <launches per pass over the batch>
"""
import csv
import sys


def main():
    path, per_pass = sys.argv[1], int(sys.argv[2])
    rows = [r for r in csv.DictReader(open(path)) if 'k_search' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    n = len(d)
    phases = [("graph replays (warm-up + timed steps)", d[:n - 2 * per_pass]),
              ("event pass, each launch alone on the GPU", d[n - 2 * per_pass:n - per_pass]),
              ("event pass, two streams overlapped", d[n - per_pass:])]
    print("search kernel launches: %d, average %.1f us" % (n, sum(d) / max(n, 1) / 1e3))
    for name, v in phases:
        print("%-45s %5d launches, average %.1f us" % (name, len(v), sum(v) / max(len(v), 1) / 1e3))


if __name__ == "__main__":
    main()
