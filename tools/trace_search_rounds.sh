# per-launch durations of k_search_round over the first plies of the bench workload (which rounds are the slow ones?)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-plies ${1:-6} > $O/ktrace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void k_search_round") or r["Kernel_Name"].startswith("k_play_move")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append("%s %.1f" % ("S" if "search" in r["Kernel_Name"] else "P", d))
print(" ".join(out))
PY
rm -rf gpurun_out/ktrace
