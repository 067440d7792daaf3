j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for m in "yolov3-rtdetr --batch 16" "yolov8s" "yolov3-tiny"; do
  for o in "" "conv_p8=2" "conv_p8=1" ""; do echo -n "[$m | $o] "; python bench.py --model $m --no-cpu-baseline --no-kernel-profile --no-parity --steps 200 --opts "$o" 2>/dev/null | j; done
done
