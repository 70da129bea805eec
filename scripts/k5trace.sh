cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k5 -o k5 -- python $GRAFT_REPO_ROOT/scripts/tune.py --batch 8 --kernel 5 --nc 3 --steps 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/k5/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pcl_" in r["Kernel_Name"]]
rows = rows[-8:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id"), r.get("Stream_Id"))
PY
