# the headline line N times in fresh processes (run-to-run spread of one build on one box): usage repeat_bench.sh N [bench args]
n=$1; shift
for i in $(seq 1 $n); do python bench.py --no-cpu-baseline --no-kernel-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'), d['config']['autotune_ms_per_step'])"; done
