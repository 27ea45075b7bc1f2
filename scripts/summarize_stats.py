import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {nsteps:g} steps -> {tot/1e6/nsteps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:100]:100s} calls={int(r['Calls'])/nsteps:8.1f}/step {float(r['TotalDurationNs'])/1e6/nsteps:8.3f} ms/step avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
