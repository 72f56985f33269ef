# round 4, first GPU call: baseline step time on this box, parity-law diagnostic, raw-input path at large batch
set -x
O=gpurun_out/r04; mkdir -p $O
python tools/bench_short.py --no-extras > $O/base.txt 2>&1
cat $O/base.txt
python tests/diag/r04_parity_law.py $O/r04_parity_law.json > $O/parity_law.log 2>&1
tail -40 $O/parity_law.log
for B in 262144 1048576; do
  for RAW in none x 1; do
    if [ $RAW = none ]; then unset DVAE_RAW_INPUTS; else export DVAE_RAW_INPUTS=$RAW; fi
    echo "B $B RAW $RAW" >> $O/bigb_raw.txt
    python tools/bench_short.py --no-extras --batch $B --steps 10 --warmup 3 --pool-gb 4 >> $O/bigb_raw.txt 2>&1
  done
done
unset DVAE_RAW_INPUTS
cat $O/bigb_raw.txt
