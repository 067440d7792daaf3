R=$PWD; mkdir -p gpurun_out/stemt; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stemt -- python3 $R/bench.py --serial --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 5 > /dev/null 2>&1
cd $R
f=$(ls -t gpurun_out/stemt/*/*kernel_stats.csv | head -1)
grep -E "stem_conv_fused|c2f16_fused" $f | cut -d, -f1-7
