cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05sf; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ks_sf16 --output-format csv -- python bench.py --workload cfg3_qcnn_sf16_b256_bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-standalone > $O/log.txt 2>&1; echo "rc=$?"
python tools/trace_step.py $O/ks_sf16_kernel_trace.csv --all > $O/r05_sf16_step_timeline.txt
mv $O/ks_sf16_kernel_stats.csv $O/r05_sf16_kernel_stats.csv
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
tail -32 $O/r05_sf16_step_timeline.txt
