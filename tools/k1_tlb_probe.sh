cd /tmp && export TMPDIR=/tmp
root=/root/repo; out=$root/gpurun_out
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat" | head -30
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" ; do
timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_k1_tlb -o k1 -- python $root/tools/bench_embed.py --uniform --vocab 4000000 --reps 3 > $out/pmc_k1_tlb.log 2>&1
python $root/tools/summarize_pmc.py $out/pmc_k1_tlb $out/r04_k1_pmc_tlb.json "rocprofv3 --pmc $set --kernel-trace -- python tools/bench_embed.py --uniform --vocab 4000000 --reps 3" | head -c 1500
done
grep -v "^[EW]2026" $out/pmc_k1_tlb.log | tail -3
