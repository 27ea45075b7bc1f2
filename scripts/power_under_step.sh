#!/bin/bash
# Diagnostic: package power and shader clock (rocm-smi) while bench.py's timed train-step loop runs (hooks = $1, e.g. "strip_pk=5:0").
cd "$(dirname "$0")/.."
UIG_DEBUG_HOOKS="$1" python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-other-configs > /tmp/_pus.json 2>/dev/null &
PID=$!
sleep 13
for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
python -c "import json; d=json.loads(open('/tmp/_pus.json').read().strip().splitlines()[-1]); print('hooks=[$1]', d['ms_per_step'], 'ms/step over', d['steps'], 'steps')"
