#!/bin/bash
# kernel trace of the pattern-compiled column kernel + stream-only kernel 3 (lab/probes/col_sparse_probe.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/col_trace; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/lab/probes/col_sparse_probe.py ${1:-8} > $O/log.txt 2>&1
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[(r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s grid %-8s lds %-7s calls %4d avg %8.1f us min %8.1f" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
P
