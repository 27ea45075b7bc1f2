"""The timed region of bench.py's `roofline` object, found again in the rocprofv3 --kernel-trace CSV of THE SAME bench.py run:
the longest run of consecutive launches of the dominant kernel with nothing else in between is bench.py's 3 x kernel_iters
back-to-back launches; its last two thirds are the launches bench.py times with HIP events.  Prints / writes their average
DURATION (kernel start -> end, no launch gaps) beside the bench line's `avg_us` (events around the region: durations + gaps).
python scripts/roofline_region_from_trace.py <kernel_trace.csv> <bench_line.json> [out.json]"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
key = "conv_strip_pk_kernel" if "strip_pk" in line["roofline"]["kernel"] else ("conv_strip_fp8" if "fp8" in line["roofline"]["kernel"] else "conv_strip_kernel")
best, cur = [], []
for r in rows:
    if key in r["Kernel_Name"] and (not cur or r["Kernel_Name"] == cur[0]["Kernel_Name"]):
        cur.append(r)
    else:
        if len(cur) > len(best):
            best = cur
        cur = [r] if key in r["Kernel_Name"] else []
if len(cur) > len(best):
    best = cur
n = len(best)
timed = best[n // 3:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
span = int(timed[-1]["End_Timestamp"]) - int(timed[0]["Start_Timestamp"])
out = {"kernel": best[0]["Kernel_Name"].split("(")[0], "run_of_consecutive_launches": n, "timed_launches": len(timed),
       "trace_avg_duration_us": round(sum(dur) / len(dur) * 1e-3, 2), "trace_span_per_launch_us": round(span / len(timed) * 1e-3, 2),
       "bench_line_avg_us": line["roofline"]["avg_us"], "bench_line_frac": line["roofline"]["frac"],
       "frac_from_trace_duration": round(line["roofline"]["flops_per_launch"] / (sum(dur) / len(dur) * 1e-9) / (line["roofline"]["peak"] * 1e12), 4),
       "note": "same bench.py process: HIP events (bench line) time durations + launch gaps under the profiler; the trace gives the durations alone"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
