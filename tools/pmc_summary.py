"""Reduce rocprofv3 --pmc output directories (counter_collection.csv) of the conv / GEMM kernels to one JSON file.

usage: python tools/pmc_summary.py <out.json> <label>=<dir>[,<dir>...] [<label>=<dir>...]

Every label names one probe (e.g. the 40x40x256 3x3 forward launch on a given kernel); its directories are the separate --pmc
passes of that probe (gpurun refuses nothing here: counters only, with --kernel-trace).  Per label the counters are averaged
over the dispatches of hconv_kernel / sgemm_kernel / igemm_kernel (the first two dispatches of a probe are warm-up and are
skipped when there are more than three) and the ratios the round's notes quote are derived:
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)     (4 SIMDs per CU; MI355X_MICROARCH.md counter units)
  issue_stall    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  parked         = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ("hconv_kernel", "sgemm_kernel", "igemm_kernel", "rowchain_fwd_kernel", "rowchain_bwd_kernel", "hwgrad_kernel", "attn_fwd_kernel", "attn_bwd_kernel")


def reduce_dirs(dirs):
    per = collections.defaultdict(list)          # counter -> [values in dispatch order]
    names = collections.Counter()
    dur = []
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in KERNELS)]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            seen = set()
            for r in rows:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    import re
                    nm = re.sub(r"\(anonymous namespace\)::|^void |bdgemm::", "", r["Kernel_Name"])
                    names[re.sub(r"\(.*", "", nm)[:110]] += 1
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {"kernels": dict(names), "dispatches_per_pass": None, "counters": {}}
    for k, v in per.items():
        v = v[2:] if len(v) > 3 else v
        out["counters"][k] = round(sum(v) / len(v), 1)
        out["dispatches_per_pass"] = len(v)
    d2 = dur[2:] if len(dur) > 3 else dur
    if d2:
        out["avg_dispatch_us_under_profiler"] = round(sum(d2) / len(d2), 2)
    c = out["counters"]
    ratio = lambda a, b, s=1.0: round(c[a] / (s * c[b]), 4) if a in c and b in c and c[b] else None
    out["derived"] = {"mfma_busy": ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", 4.0), "issue_stall": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                      "parked": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "valu_per_mfma": ratio("SQ_INSTS_VALU", "SQ_INSTS_MFMA"),
                      "lds_conflict": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
    return out


def main():
    out_path, res = sys.argv[1], {}
    for spec in sys.argv[2:]:
        label, dirs = spec.split("=", 1)
        res[label] = reduce_dirs(dirs.split(","))
        print(label, res[label]["derived"], res[label].get("avg_dispatch_us_under_profiler"))
    meta = {"source": "rocprofv3 --pmc <counters> --kernel-trace (counters only; separate passes per counter group), tools/p16_pmc_probe.py",
            "commit": os.environ.get("BDETR_COMMIT") or os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip() or None}
    json.dump({"meta": meta, "probes": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
