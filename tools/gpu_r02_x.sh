# PMC counters of the encoder attention kernel (what bounds it?): three passes of a short single-lane bench run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmcx; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmcx
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVES" "SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --lanes 1 --device-batch 8 --fixed-steps 4 > $OUT/b$i.log 2> $OUT/b$i.err )
done
ls $OUT | head
python tools/pmc_kernel_counters.py enc_attn_lds $OUT/p1_results.db $OUT/p2_results.db $OUT/p3_results.db
python tools/pmc_kernel_counters.py gemm256_kernelIDF16_Li1 $OUT/p1_results.db $OUT/p2_results.db $OUT/p3_results.db
rm -f $OUT/*.db
