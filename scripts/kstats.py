"""Print a rocprofv3 kernel_stats.csv compactly: kernel (short name), calls, average / min / max microseconds."""
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", row["Name"])
    name = re.sub(r"\(.*", "", name)
    print(f"{name[:48]:48s} {int(row['Calls']):6d}  avg {float(row['AverageNs']) / 1e3:9.2f} us  min {float(row['MinNs']) / 1e3:9.2f}  max {float(row['MaxNs']) / 1e3:9.2f}  total {float(row['TotalDurationNs']) / 1e6:9.3f} ms")
