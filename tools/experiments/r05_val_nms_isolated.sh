R=$PWD; mkdir -p gpurun_out/iso; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iso -- python3 $R/tools/experiments/r05_val_nms_counts.py --batches 8 > /dev/null 2>&1
cd $R
f=$(ls -t gpurun_out/iso/*/*kernel_stats.csv | head -1)
grep -i "nms\|zero_words" $f | cut -d, -f1-7 | cut -c1-60,100-400
