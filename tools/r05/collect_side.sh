#!/bin/bash
# Round-5 evidence for the non-train kernels (STFT / ISTFT / MCEM): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE (own passes) +
# an SQ pass for the MCEM chain kernel.  Run on the GPU box; outputs in gpurun_out/r05c/.
O=$PWD/gpurun_out/r05c; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for what in stft mcem; do
  if [ $what = stft ]; then CMD="python3 $R/tools/bench_stft.py"; else CMD="python3 $R/tools/bench_mcem.py --no-cpu --batch 8 25"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${what}_stats -- $CMD > $O/${what}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${what}_fetch -- $CMD > $O/${what}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${what}_write -- $CMD > $O/${what}_write.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mcem_sq -- python3 $R/tools/bench_mcem.py --no-cpu --batch 8 25 > $O/mcem_sq.log 2>&1
cd $R
cp $O/stft_stats/*/*kernel_stats.csv $O/stft_kernel_stats.csv 2>/dev/null
cp $O/mcem_stats/*/*kernel_stats.csv $O/mcem_kernel_stats.csv 2>/dev/null
python tools/r03/side_summary.py $O > $O/side_summary.json 2> $O/side_summary.err
tail -c 3000 $O/side_summary.json
# keep the merge-back small (gpurun merges at most 64 MiB): drop the raw per-dispatch traces and counter tables once summarised
find $O \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.db" \) -delete
