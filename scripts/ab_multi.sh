# A/B several builds of libuig.so (ab/*.so) on the same box.  usage: bash scripts/ab_multi.sh <python script> [args]
for rep in 1 2; do
  for l in ab/*.so; do
    echo "== $l"; UIG_LIB_PATH=$PWD/$l timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
