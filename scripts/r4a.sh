source scripts/r3_run.sh r4a
export PK_VARIANTS="pk,pk 4iss,pk 4iss-u"
step pk 600 python scripts/bench_strip_pk.py
tail -16 gpurun_out/r4a_pk.log; tail -3 gpurun_out/r4a_pk.err
