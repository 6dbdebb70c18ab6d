#!/bin/bash
# rocprofv3 kernel trace of the default bench step -> gpurun_out/r04/{r04_qcnn_bf16_b256_kernel_stats.csv, r04_qcnn_step_timeline.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r04; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ks_qcnn --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-standalone "$@" > $O/log_qcnn.txt 2>&1; echo "qcnn trace rc=$?"
python tools/trace_step.py $O/ks_qcnn_kernel_trace.csv --all > $O/r04_qcnn_step_timeline.txt
mv $O/ks_qcnn_kernel_stats.csv $O/r04_qcnn_bf16_b256_kernel_stats.csv
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
tail -20 $O/r04_qcnn_step_timeline.txt
