# dev tool (GPU box): compute-side PMC counters of the encoder kernels of THIS round (the round-2 file profiles/r02_x_pmc_attention_fc1.txt predates the
# k64 GEMM and was never refreshed): three separate --pmc passes over a short single-lane run of the bench command, raw per-launch averages per kernel.
#   gpurun --timeout 900 -- 'bash tools/gpu_r05_pmc_compute.sh'
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/pmc_compute && OUT=$PWD/gpurun_out/pmc_compute
export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  i=$((i + 1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --no-cpu-baseline --headline-only --fixed-steps 4 --inflight 1 --lanes 1 > $OUT/b$i.log 2> $OUT/b$i.err )
done
python - <<'PY' > gpurun_out/r05_ag_pmc_compute.txt
import sqlite3, glob, collections
per = collections.defaultdict(dict)
for db in sorted(glob.glob("gpurun_out/pmc_compute/p*_results.db")):
    c = sqlite3.connect(db)
    for name, cn, n, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        per[name][cn] = (n, tot)
keep = ("enc_attn", "gemm256k64", "layernorm_kernel", "dec_cross_attn", "dec_gemv_kernel", "dec_reduce_ln", "dec_self_attn")
print("per-launch averages, `bench.py --steps 3 --fixed-steps 4 --inflight 1 --lanes 1` (one lane: kernels run alone), three separate --pmc passes\n")
for name in sorted(per, key=lambda k: -per[k].get("SQ_BUSY_CYCLES", (0, 0))[1]):
    if not any(k in name for k in keep):
        continue
    v = per[name]
    n = max(x[0] for x in v.values())
    print(name[:110])
    for cn in sorted(v):
        print(f"   {cn:28s} launches {v[cn][0]:6d}  per launch {v[cn][1] / max(1, v[cn][0]):.4g}")
    g = lambda k: v.get(k, (1, 0.0))[1] / max(1, v.get(k, (1, 0.0))[0])
    if g("SQ_INSTS_MFMA") > 0:
        print(f"   -> VALU (non-MFMA) instructions per MFMA: {(g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'):.2f};  transcendental per MFMA: {g('SQ_INSTS_VALU_TRANS_F32') / g('SQ_INSTS_MFMA'):.2f};"
              f"  SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs... raw ratio) {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1.0, g('SQ_BUSY_CYCLES')):.3f}")
    print()
PY
rm -f gpurun_out/pmc_compute/*.db
head -80 gpurun_out/r05_ag_pmc_compute.txt | cut -c1-200
