"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace, CSV output)
into HBM bytes per igemm launch and per step, with the gfx950 corrections of MI355X_MICROARCH.md
(FETCH_SIZE counts 128-B requests at 64 B -> x2; both counters are in KB).

usage: python tools/hbm_traffic.py <fetch_dir> <write_dir> <steps_in_trace> <out.json>"""
import csv, glob, json, os, sys


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


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = counter_sums(fetch_dir, "FETCH_SIZE"), counter_sums(write_dir, "WRITE_SIZE")
    ig = lambda d: [v for k, v in d.items() if "igemm_kernel" in k]
    n = sum(v[0] for v in ig(fetch))
    fb = sum(v[1] for v in ig(fetch)) * 2 * 1024
    wb = sum(v[1] for v in ig(write)) * 1024
    allb = sum(v[1] for v in fetch.values()) * 2 * 1024 + sum(v[1] for v in write.values()) * 1024
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py, igemm_kernel dispatches only",
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), units KB -> x1024 (MI355X_MICROARCH.md, HBM section)",
           "steps_in_trace": steps, "igemm_launches": n,
           "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "hbm_bytes_per_launch": (fb + wb) / n,
           "igemm_hbm_gb_per_step": (fb + wb) / steps / 1e9, "all_kernels_hbm_gb_per_step": allb / steps / 1e9}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
