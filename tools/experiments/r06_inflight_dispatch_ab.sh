j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
B="python bench.py --no-cpu-baseline --no-kernel-profile --no-parity"
for o in "" "c2f=6" "c2f=0" "conv_ws3=0" "c2f_stream_rows=40" "" "c2f=6" "c2f=0"; do echo -n "[$o] "; $B --opts "$o" 2>/dev/null | j; done
