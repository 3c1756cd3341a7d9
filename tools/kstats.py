"""Prints name (shortened), calls, average / min ns per kernel from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        n = row["Name"]
        if "multi_kernel" in n or "llk_" in n:
            print("%-50s calls %5s avg %9.0f ns min %9s" % (n.split("(")[0][-48:], row["Calls"], float(row["AverageNs"]), row["MinNs"]))
