"""Per-kernel cost of ONE train step from a rocprofv3 --kernel-trace CSV (e.g. the serialised trace of scripts/trace_serial.sh):
kernel name -> launches, total ms, share of the step; plus coarse families.  python scripts/summarize_trace.py <kernel_trace.csv> [out.csv]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n)
    r["n"] = n.split("(")[0][:90]
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_flat" in r["n"]]
# two Adam launches per step (G, D): a step window = from the end of the D Adam two steps back to the end of the last D Adam
t1, t0 = rows[adam[-1]]["e"], rows[adam[-3]]["e"]
win = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
tot = sum(r["e"] - r["s"] for r in win)
byn = collections.defaultdict(lambda: [0, 0])
for r in win:
    byn[r["n"]][0] += 1; byn[r["n"]][1] += r["e"] - r["s"]
fam = collections.Counter()
def family(n):
    for key, f in (("conv_strip", "strip conv"), ("wgrad", "weight gradients"), ("in_", "InstanceNorm"), ("igemm", "generic gather conv"),
                   ("conv_tr2", "transposed stride-2 conv"), ("rowstrip", "7x7 layers"), ("headrow", "7x7 layers"), ("cin8", "7x7 layers"),
                   ("gemv", "PatchGAN head"), ("adam", "Adam"), ("pack", "weight packing"), ("loss", "losses")):
        if key in n:
            return f
    return "other pointwise / layout"
print(f"step window {1e-6 * (t1 - t0):.3f} ms wall, {len(win)} kernels, sum of durations {1e-6 * tot:.3f} ms")
out = []
for n, (c, d) in sorted(byn.items(), key=lambda kv: -kv[1][1]):
    fam[family(n)] += d
    out.append((n, c, d * 1e-6, d / c * 1e-3, 100.0 * d / tot))
for n, c, ms, us, pct in out[:28]:
    print(f"{ms:8.3f} ms {pct:5.1f} %  x{c:4d}  {us:8.1f} us  {n}")
print("families:")
for f, d in fam.most_common():
    print(f"  {f:28s} {1e-6 * d:7.3f} ms  {100.0 * d / tot:5.1f} %")
# the gather / transposed / generic weight-gradient launches one by one, in step order (which layer costs what): name, grid, duration
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fo:
        w = csv.writer(fo); w.writerow(["order", "kernel", "grid_x", "grid_y", "grid_z", "wg_x", "lds_bytes", "us"])
        for i, r in enumerate(win):
            if any(k in r["n"] for k in ("igemm", "conv_tr2", "wgrad_kernel", "cin8", "headrow", "rowstrip", "gemv", "wgrad_head")):
                w.writerow([i, r["n"], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""),
                            r.get("LDS_Block_Size", r.get("Group_Segment_Size", "")), round((r["e"] - r["s"]) * 1e-3, 2)])
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fo:
        w = csv.writer(fo); w.writerow(["kernel", "launches_per_step", "total_ms", "avg_us", "pct_of_step"])
        w.writerows(out)
