# A/B two builds of libuig.so on the same box: ab/libuig_base.so (reference build) vs the in-tree library.
# usage: bash scripts/ab.sh <python script> [args]
for rep in 1 2; do
  echo "== base"; UIG_LIB_PATH=$PWD/ab/libuig_base.so timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  echo "== new";  timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
