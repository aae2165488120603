#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of config 3's deep layers in isolation, per forced F(2x2,3x3) tile configuration - the Python
# loop of tools/bench_small_maps.py is host-bound at these sizes, so its own event timings say nothing: read the trace.
#   tools/small_maps_trace.sh <outdir> "7 8 9"
OUT=$(realpath ${1:-gpurun_out/small_maps}); mkdir -p $OUT
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for k in ${2:-7 8 9}; do
  export FORCE_WKIND=$k
  rocprofv3 --kernel-trace --output-format csv -d $OUT/k$k -o t -- python3 $REPO/tools/bench_small_maps.py > $OUT/k$k.log 2>&1
done
cd $REPO
python3 - $OUT ${2:-7 8 9} <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for k in sys.argv[2:]:
    f = glob.glob(os.path.join(out, "k" + k, "**", "*kernel_trace.csv"), recursive=True)[0]
    seq = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "wino2_kernel" in n or "wino_kernel" in n or "splitk_finish" in n:
            seq.append((n[:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    agg = collections.OrderedDict()
    for n, g, us in seq:
        agg.setdefault((n, g), []).append(us)
    print("== forced configuration", k)
    for (n, g), v in agg.items():
        v.sort()
        print("   %-62s %5d workgroups  %4d launches  median %7.1f us" % (n, g, len(v), v[len(v) // 2]))
PY
