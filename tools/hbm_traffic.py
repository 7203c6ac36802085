"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace, CSV output)
into HBM bytes per GEMM-family launch (igemm_kernel + sgemm_kernel + hconv_kernel), per step, and per kernel name, with the gfx950
corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B -> x2; both counters are in KB).

usage: python tools/hbm_traffic.py <fetch_dir> <write_dir> <steps_in_trace> <out.json> [kernel_stats.csv <steps_in_stats_trace>]
With the optional kernel-stats CSV (rocprofv3 --kernel-trace --stats of the same command) every kernel also gets its
average HBM rate = bytes per step / time per step.  BDETR_COMMIT (the commit the snapshot was taken from; the GPU box has no .git)
is recorded as `profiled_commit`."""
import csv, glob, json, os, re, sys


def counter_sums(d, counter):
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = per_kernel.setdefault(r["Kernel_Name"], [0, 0.0])
            k[0] += 1
            k[1] += float(r["Counter_Value"])
    return per_kernel


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void |bdgemm::", "", name)
    return re.sub(r"\(.*", "", name)[:120]


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = counter_sums(fetch_dir, "FETCH_SIZE"), counter_sums(write_dir, "WRITE_SIZE")
    fam = lambda d: [v for k, v in d.items() if "igemm_kernel" in k or "sgemm_kernel" in k or "hconv_kernel" in k]
    n = sum(v[0] for v in fam(fetch))
    fb = sum(v[1] for v in fam(fetch)) * 2 * 1024
    wb = sum(v[1] for v in fam(write)) * 1024
    allb = sum(v[1] for v in fetch.values()) * 2 * 1024 + sum(v[1] for v in write.values()) * 1024
    times = {}
    if len(sys.argv) > 6:
        tsteps = float(sys.argv[6])
        for r in csv.DictReader(open(sys.argv[5])):
            times[r["Name"]] = float(r["TotalDurationNs"]) / tsteps * steps          # scaled to this trace's step count
    per_kernel = []
    for k in set(fetch) | set(write):
        f_ = fetch.get(k, [0, 0.0]); w_ = write.get(k, [0, 0.0])
        b = f_[1] * 2 * 1024 + w_[1] * 1024
        row = {"kernel": short(k), "launches_per_step": round(max(f_[0], w_[0]) / steps, 1), "read_gb_per_step": round(f_[1] * 2 * 1024 / steps / 1e9, 3),
               "write_gb_per_step": round(w_[1] * 1024 / steps / 1e9, 3)}
        if k in times and times[k] > 0:
            row["ms_per_step"] = round(times[k] / steps / 1e6, 3)
            row["tb_per_s"] = round(b / times[k] / 1e3, 2)
        per_kernel.append((b, row))
    per_kernel.sort(key=lambda t: -t[0])
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; GEMM family = igemm_kernel + sgemm_kernel + hconv_kernel dispatches",
           "profiled_commit": os.environ.get("BDETR_COMMIT"), "steps": steps,
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), units KB -> x1024 (MI355X_MICROARCH.md, HBM section)",
           "steps_in_trace": steps, "igemm_launches": n,
           "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "hbm_bytes_per_launch": (fb + wb) / n,
           "igemm_hbm_gb_per_step": (fb + wb) / steps / 1e9, "all_kernels_hbm_gb_per_step": allb / steps / 1e9,
           "per_kernel": [r for _, r in per_kernel[:24]]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))
    for r in res["per_kernel"][:14]:
        print(r)


if __name__ == "__main__":
    main()
