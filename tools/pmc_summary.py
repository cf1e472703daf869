"""Summarise rocprofv3 --pmc passes of one kernel into profiles/<name>.json.

Each pass is its own rocprofv3 run (one counter group per pass; gpurun refuses --pmc together with the
trace domains), e.g. on the GPU box:
   for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
     rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_r02/<tag> -- python3 tools/one_gemm.py 21600 3072 768 2 19 20
   done
   python tools/pmc_summary.py gpurun_out/pmc_r02 gemm256p_kernel profiles/r02_dominant_kernel_pmc.json \
          --family gemm_bf16_gelu_256x256pp_n3072k768 --streams 30 --algorithmic-bytes 170590208

Counter values are averaged over the dispatches of the kernel whose name contains the given substring.
traffic_bytes_per_launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB: on gfx950 FETCH_SIZE reports exactly half
of the bytes of wide coalesced reads (MI355X_MICROARCH.md, section HBM), WRITE_SIZE is exact for 16-B stores."""
import argparse
import csv
import glob
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("kernel_substr")
ap.add_argument("out")
ap.add_argument("--family", default="")
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--algorithmic-bytes", type=float, default=0.0)
ap.add_argument("--command", default="")
ap.add_argument("--kernel-sha", default="", help="vt_build_info()'s k_gemm256 of the library the passes ran on (bench.py prints the traffic only for that build)")
a = ap.parse_args()

acc, cnt, dur = {}, {}, []
for f in glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if a.kernel_substr not in row["Kernel_Name"]:
                continue
            k = row["Counter_Name"]
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
            if k == "GRBM_GUI_ACTIVE":
                dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
if not acc:
    sys.exit(f"no dispatches of a kernel containing '{a.kernel_substr}' under {a.root}")
avg = {k: acc[k] / cnt[k] for k in sorted(acc)}
out = {"kernel": a.kernel_substr, "kernel_family": a.family, "streams_per_pass": a.streams,
       "dispatches_averaged": max(cnt.values()), "counters": avg}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    rd, wr = 2.0 * avg["FETCH_SIZE"] * 1024.0, avg["WRITE_SIZE"] * 1024.0
    out["fabric_read_bytes_per_launch"] = rd
    out["write_bytes_per_launch"] = wr
    out["traffic_bytes_per_launch"] = rd + wr
    if a.algorithmic_bytes:
        out["algorithmic_bytes_per_launch"] = a.algorithmic_bytes
        out["traffic_over_algorithmic"] = (rd + wr) / a.algorithmic_bytes
if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
    out["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
if dur:
    out["kernel_ns_under_pmc"] = sum(dur) / len(dur)
out["provenance"] = ("rocprofv3 --pmc passes (one counter group per run) of `" + (a.command or "tools/one_gemm.py") +
                     "` on MI355X, summarised by tools/pmc_summary.py; reads = 2 x FETCH_SIZE (gfx950 "
                     "correction, MI355X_MICROARCH.md section HBM), writes = WRITE_SIZE")
if a.kernel_sha:
    out["kernel_source_sha256"] = a.kernel_sha
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))
