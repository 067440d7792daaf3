# same-box A/B of libraries (UPA_HIP_LIB) on the headline line, order-balanced: usage ab_libs.sh libA.so libB.so ...
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
rev=(); for t in "$@"; do rev=("$t" "${rev[@]}"); done
for l in "$@" "${rev[@]}"; do echo "LIB $l"; UPA_HIP_LIB=$PWD/ultralytics_pro_amd/$l python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | j; done
