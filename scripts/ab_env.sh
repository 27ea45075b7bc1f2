# A/B an environment toggle on one box: bash scripts/ab_env.sh UIG_PARALLEL_BACKWARD=0 -- bench.py --no-cpu-baseline
E="$1"; shift; shift
for rep in 1 2 3; do
  echo -n "default: "; timeout -k 10 300 python "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  echo -n "$E: "; env "$E" timeout -k 10 300 python "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
