# A/B a hook on one box, printing step throughput and the dominant kernel's HIP-event time: bash scripts/ab_roof.sh strip=3
for rep in 1 2 3; do
  for H in "" "$1"; do
    UIG_DEBUG_HOOKS="$H" timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hooks=[$H]', d['value'], 'img/s  dominant', d['roofline']['avg_us'], 'us  frac', d['roofline']['frac'])" || exit 1
  done
done
