#!/bin/bash
# HBM bytes of the column-group Hessian kernel (order 8, 8 trajectories per launch) per option set: separate WRITE_SIZE / FETCH_SIZE passes.
# usage: hess_traffic.sh <tag> [key=value ...]   -> gpurun_out/r05/hess_traffic_<tag>.txt
ROOT=$(pwd); TAG=$1; shift; OUT=$ROOT/gpurun_out/r05/hess_pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for P in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/$P -o hc -- python $ROOT/lab/probes/hess_cols_run.py 8 8 8 "$@" > /dev/null 2>&1
done
cd $ROOT
python - "$TAG" <<'PY' > gpurun_out/r05/hess_traffic_$TAG.txt
import csv, glob, collections, sys
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/r05/hess_pmc_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hess_cols' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
alg = 8 * 99 * 20440 * 8 / 1e6
for k in sorted(tot):
    v = tot[k] / max(n[k], 1)
    mb = v * 1024 / 1e6 * (2.0 if k == 'FETCH_SIZE' else 1.0)  # KB units; FETCH_SIZE doubled (gfx950 note of the guide, as profiles/pmc_traffic.json)
    print("%s %s: %.1f per dispatch over %d dispatches = %.1f MB%s" % (sys.argv[1], k, v, n[k], mb, (" = %.3f x the algorithmic %.1f MB" % (mb / alg, alg)) if k == 'WRITE_SIZE' else ""))
PY
cat gpurun_out/r05/hess_traffic_$TAG.txt
