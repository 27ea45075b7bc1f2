# A/B several kernel-selection hook settings on one box: bash scripts/ab_hooks_multi.sh "strip=5" "strip=7" -- bench.py --no-cpu-baseline
HOOKS=(); while [ "$1" != "--" ]; do HOOKS+=("$1"); shift; done; shift
for rep in 1 2 3; do
  echo "== default"; timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110 || exit 1
  for H in "${HOOKS[@]}"; do
    echo "== hooks $H"; UIG_DEBUG_HOOKS="$H" timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110 || exit 1
  done
done
