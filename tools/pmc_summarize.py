#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE per kernel family from the rocprofv3 --pmc CSVs written by
tools/pmc_traffic.sh and apply the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact.
Counter unit: KiB (rocprofv3 derived counter, TCC_EA0_*REQ based).
usage: pmc_summarize.py <outdir> [pairs that went through the kernels, default 3]; figures are per frame pair."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if row.get("Counter_Name") != ctr:
                continue
            fam = next((k for k in ("wino_pack_kernel", "wino4_pack_kernel", "wino1d_pack_kernel", "pack32_batch_kernel", "wino7_pack_kernel", "wino5_pack_kernel", "wino7s_kernel", "wino7_kernel", "wino5s_kernel", "wino5_kernel", "wino4_kernel", "wino1d_kernel", "wino2_kernel", "wino_kernel", "final_conv_valu_kernel", "final_conv_kernel", "conv16_ups_kernel", "conv16_multi_kernel", "conv16_kernel", "conv_mfma_kernel", "upsample2x_cat_hl8_kernel", "upsample2x_cat_kernel",
                                    "flowinterp_inputs_hl8_kernel", "flowinterp_inputs_kernel", "to_hl8_kernel", "to_hq8_kernel", "gather_cols_kernel", "pack16_kernel",
                                    "synthesize_kernel", "copy_view_kernel", "pack_weights_kernel", "FillFunctor",
                                    "copyBuffer") if k in name), "other")
            res[fam][ctr] += float(row["Counter_Value"])
            if ctr == "FETCH_SIZE":
                calls[fam] += 1
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3   # frame pairs processed (bench.py --steps 2 --warmup 1 x pairs per step)
summ = {}
for fam, d in res.items():
    rd = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0      # gfx950: x2
    wr = d.get("WRITE_SIZE", 0.0) * 1024.0
    summ[fam] = {"launches": calls[fam], "hbm_read_bytes_per_step": rd / STEPS, "hbm_write_bytes_per_step": wr / STEPS,
                 "hbm_bytes_per_step": (rd + wr) / STEPS}
json.dump(summ, open(os.path.join(out, "pmc_traffic_summary.json"), "w"), indent=1)
for fam, v in sorted(summ.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:12]:
    print("%-42s launches %4d  read %8.1f MB  write %8.1f MB per step" % (fam, v["launches"], v["hbm_read_bytes_per_step"] / 1e6, v["hbm_write_bytes_per_step"] / 1e6))
