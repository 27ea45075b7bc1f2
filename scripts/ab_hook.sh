# A/B one kernel-selection hook on the same box and library: bash scripts/ab_hook.sh "tr2=0" bench.py --no-cpu-baseline
H="$1"; shift
for rep in 1 2; do
  echo "== hooks $H"; UIG_DEBUG_HOOKS="$H" timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  echo "== default"; timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
