"""Print the head of a rocprofv3 kernel_stats.csv with short kernel names (profiling helper)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.1f} ms over {len(rows)} kernels")
for r in rows[:n]:
    name = r["Name"].replace("tsamd::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    print(f"{name:66s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {int(r['TotalDurationNs']) / 1e6:8.1f} ms  {float(r['Percentage']):5.1f} %")
