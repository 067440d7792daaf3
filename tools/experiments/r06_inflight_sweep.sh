# same-box sweep of the throughput runner's shape on the headline line: steps in flight x resident input batches (round 6)
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('steps_in_flight'), d['config'].get('micro_batches'))"; }
B="python bench.py --no-cpu-baseline --no-kernel-profile --no-parity"
echo "autotune"; $B 2>/dev/null | j
for f in 3 4 5 6 8; do echo "in-flight $f linear"; $B --in-flight $f --micro-batches 1 2>/dev/null | j; done
for f in 4 6; do echo "in-flight $f, 16 input batches"; $B --in-flight $f --micro-batches 1 --input-batches 16 2>/dev/null | j; done
echo "autotune again"; $B 2>/dev/null | j
