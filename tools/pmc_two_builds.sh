set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05/pmc_rep; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  if [ $v = b ]; then export SALNMF_LIB=$REPO/salamander_amd/lib/libsalnmf_b.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_trace -- python3 $REPO/tools/time_mv_c4.py > $OUT/${v}_trace.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${v}_pmc1 -- python3 $REPO/tools/time_mv_c4.py > $OUT/${v}_pmc1.log 2>&1
  ( cd $REPO && python3 tools/pmc_summary.py $OUT/${v}_summary.json $OUT/${v}_trace $OUT/${v}_pmc1 > $OUT/${v}_summary.txt 2>&1 )
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
done
head -6 $OUT/a_summary.txt; head -6 $OUT/b_summary.txt
