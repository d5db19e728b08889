"""Print name prefix / calls / avg us from a rocprofv3 kernel_stats.csv (development tool)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s %6s calls %10.1f us avg" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
