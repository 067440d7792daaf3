timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "detect" > gpurun_out/r06_t4.log 2>&1; echo rc=$? >> gpurun_out/r06_t4.log
B="python bench.py --no-cpu-baseline --no-parity --no-kernel-profile"
for o in "" "detect_stream=2" "detect_stream=2,detect_stream_rows=40"; do
  $B --opts "$o" > gpurun_out/r06_ab_inflight_"$o".log 2>&1
  $B --serial --opts "$o" > gpurun_out/r06_ab_serial_"$o".log 2>&1
done
for f in gpurun_out/r06_ab_*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done > gpurun_out/r06_ab_summary.txt 2>&1
timeout 1500 python -m pytest tests/test_hip_e2e.py -m gpu -q -x > gpurun_out/r06_t5.log 2>&1; echo rc=$? >> gpurun_out/r06_t5.log
