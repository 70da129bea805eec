#!/usr/bin/env python3
"""Average per dispatch of every counter in rocprofv3 counter_collection CSVs, for kernels whose name contains argv[2]."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
