"""One batch-1 generator inference (Translator, eager launches so that every kernel shows in a rocprofv3 --kernel-trace) repeated a few
times; summarise with: python scripts/trace_infer.py summarize <kernel_trace.csv>"""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    import csv, collections, re
    rows = list(csv.DictReader(open(sys.argv[2])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:80]
    rows.sort(key=lambda r: r["s"])
    # the last run = the kernels after the last long gap
    cut = max(range(1, len(rows)), key=lambda i: (rows[i]["s"] - rows[i - 1]["e"]) if i > len(rows) // 2 else -1)
    win = rows[cut:]
    tot = sum(r["e"] - r["s"] for r in win)
    print(f"last run: {len(win)} kernels, wall {(win[-1]['e'] - win[0]['s']) / 1e3:.1f} us, sum of durations {tot / 1e3:.1f} us")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in win:
        agg[r["n"]][0] += 1; agg[r["n"]][1] += r["e"] - r["s"]
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{d / 1e3:8.1f} us  x{c:3d}  {d / c / 1e3:7.1f} us  {n}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd.inference import Translator
g = u.Generator(n_blocks=9, dtype=torch.bfloat16)
x = torch.rand(1, 256, 256, 8, device="cuda").to(torch.bfloat16)
tr = Translator(g, use_graph=False)
for _ in range(5):
    tr.run_phys(x)
    torch.cuda.synchronize(); time.sleep(0.05)
