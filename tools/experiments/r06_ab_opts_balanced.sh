# order-balanced same-box A/B of upa_opts settings on the headline line: usage r06_ab_opts_balanced.sh "optsA" "optsB" [rounds]
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
B="python bench.py --no-cpu-baseline --no-kernel-profile --no-parity"
n=${3:-3}
for r in $(seq 1 $n); do for o in "$1" "$2" "$2" "$1"; do echo -n "[$o] "; $B --opts "$o" 2>/dev/null | j; done; done
