"""rocprofv3 --kernel-trace CSV -> one row per (kernel, grid size): launches and average / total duration.  The per-kernel --stats
table hides which LAYERS of a kernel are the slow ones; the grid size identifies the layer."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:90]


def main(path, steps):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y") or "", r.get("Grid_Size_Z") or "")
        a = agg[(short(r["Kernel_Name"]), g)]
        a[0] += 1; a[1] += d
    print("kernel,grid_x,grid_y,grid_z,launches_per_step,avg_us,ms_per_step")
    for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{g[0]},{g[1]},{g[2]},{n / steps:.2f},{t / n:.1f},{t / 1e3 / steps:.4f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
