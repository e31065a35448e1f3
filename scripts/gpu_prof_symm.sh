# kernel split of the class-shared symmetric apply (k_fxs_symm8 / k_fxs_symfin) at configs[2] size, and the segment-length sweep
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_symm -- python3 $R/scripts/symv_tune.py class_sym > $R/gpurun_out/prof_symm.txt 2>&1
f=$(find $R/gpurun_out/prof_symm -name "*kernel_stats.csv" | head -n 1)
cut -d, -f1-4 $f | cut -c1-150 | sed -n 1,8p
find $R/gpurun_out/prof_symm -name "*kernel_trace.csv" -delete
