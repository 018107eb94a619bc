"""Per-step kernel table from two rocprofv3 --kernel-trace --stats runs of the SAME command with different step counts (what is left
after the difference is the replayed step: capture-time and set-up launches cancel).
usage: python tools/diag/step_kernels.py A_kernel_stats.csv B_kernel_stats.csv steps_B_minus_steps_A [top]"""
import csv
import sys


def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


a, b, dsteps = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = []
for k, (c1, t1) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    dc, dt = (c1 - c0) / dsteps, (t1 - t0) / dsteps / 1e6
    if dc > 0:
        rows.append((dt, dc, k))
rows.sort(reverse=True)
print(f"total {sum(r[0] for r in rows):.3f} ms/step in {sum(r[1] for r in rows):.0f} launches; "
      f"launches under 12 us: {sum(r[1] for r in rows if r[0] / r[1] * 1e3 < 12):.0f} = {sum(r[0] for r in rows if r[0] / r[1] * 1e3 < 12):.3f} ms")
for dt, dc, k in rows[:top]:
    print(f"{dt:8.3f} ms {dc:7.1f} calls {dt / dc * 1e3:9.1f} us  {k[:120]}")
