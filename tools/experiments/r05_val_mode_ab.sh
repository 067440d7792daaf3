run() { python bench.py --workload val "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('dispatch_by_mode'), d['config'].get('dispatch_opts'))"; }
for r in 1 2 3; do
run --no-mode-dispatch
run
run --opts c2f=4
run --opts c2f_stream_rows=-1
run --opts conv_ws3=1
done
