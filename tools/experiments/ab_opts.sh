# same-box A/B of dispatch options on the headline line: usage ab_opts.sh "<bench args A>" "<bench args B>" ... (order, then reversed)
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
rev=(); for t in "$@"; do rev=("$t" "${rev[@]}"); done
for a in "$@" "${rev[@]}"; do echo "ARGS $a"; python bench.py --no-cpu-baseline --no-kernel-profile $a 2>/dev/null | j; done
