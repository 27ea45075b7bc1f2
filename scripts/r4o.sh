source scripts/r3_run.sh r4o
step smoke 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')"
step bench 600 python bench.py
tail -3 gpurun_out/r4o_smoke.log
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4o_bench.log").read().strip().splitlines()[-1]); print("ms/step", j["ms_per_step"], "img/s", j["value"], "roofline", j["roofline"]["avg_us"], j["roofline"]["frac"], j["roofline"]["in_step_frac"], "g_fwd", j["g_fwd"]["mfma_frac"]); print({k:(v["ms_per_step"],v["images_per_s"]) for k,v in j["other_configs"].items()}); print(j["cpu_baseline"])
PY
