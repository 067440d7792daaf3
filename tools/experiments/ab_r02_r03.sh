# same-box A/B of builds checked out as worktrees under the repo root (e.g. `git worktree add .r02_tree 79fe3d3 && make -C
# .r02_tree/ultralytics_pro_amd/csrc -j6`; usage: ab_r02_r03.sh ".|" ".r02_tree|" - remove the worktree afterwards) (they travel with the gpurun snapshot): headline line of each
# "dir|extra bench args", in the order given and then reversed (runs later in a call measure a warmer, slower chip)
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
rev=(); for t in "$@"; do rev=("$t" "${rev[@]}"); done
for t in "$@" "${rev[@]}"; do d=${t%%|*}; a=${t#*|}; echo "$t"; (cd $d && python bench.py --no-cpu-baseline --no-kernel-profile $a 2>/dev/null | j); done
