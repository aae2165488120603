#!/bin/bash
# F(2x2,3x3) (+ split-K) against F(4x4,3x3) per 3x3 layer of the U-Net at a training shape, from kernel-trace durations (the Python loop is
# host-bound at these sizes): tools/wino24_trace.sh <outdir> [B] [H] [W]
OUT=$(realpath ${1:-gpurun_out/wino24}); mkdir -p $OUT
B=${2:-2}; H=${3:-352}; W=${4:-352}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
export NO_DIRECT=1
W4=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/w2 -o t -- python3 $REPO/tools/bench_layers_wino.py $B $H $W > $OUT/w2.log 2>&1
W4=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/w4 -o t -- python3 $REPO/tools/bench_layers_wino.py $B $H $W > $OUT/w4.log 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys, re
out = sys.argv[1]
def layers(tag):
    names = [l.split()[0] for l in open(os.path.join(out, tag + ".log")) if re.match(r"^(conv|fuse)", l) and "skipped" not in l]
    f = glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if re.search(r"wino\d?_kernel|wino_kernel|splitk_finish", n):
            seq.append((n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    # every layer: 7 calls (2 warm-up + 5 timed), each call = the conv launch (+ a finish launch when split; sub-pixel: + border)
    per, i = {}, 0
    for name in names:
        first = seq[i][0]
        j = i + 1
        while j < len(seq) and seq[j][0] != first:
            j += 1
        k = j - i                      # launches per call
        calls = seq[i:i + 7 * k]
        tot = sorted(sum(c[2] for c in calls[q * k:(q + 1) * k]) for q in range(7))
        per[name] = (tot[3], k, calls[0][1], re.sub(r"void |\(.*", "", calls[0][0])[:44])
        i += 7 * k
    return per
p2, p4 = layers("w2"), layers("w4")
print("%-10s | %9s %3s %6s %-44s | %9s %3s %6s %-44s | %s" % ("layer", "F(2x2) us", "n", "wgs", "kernel", "F(4x4) us", "n", "wgs", "kernel", "faster"))
for name in p2:
    a, b = p2[name], p4.get(name)
    if b is None:
        continue
    print("%-10s | %9.1f %3d %6d %-44s | %9.1f %3d %6d %-44s | %s" % (name, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], "F(2x2) %.2fx" % (b[0] / a[0]) if a[0] < b[0] else "F(4x4) %.2fx" % (a[0] / b[0])))
PY
