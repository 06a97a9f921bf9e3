#!/bin/bash
# round 6: what bounds amis_backward_mfma_kernel at C2?  Launch shapes (waves x resident tiles -> waves per SIMD) for the current and the
# previous pair loop, then the SQ counters of both.   -> gpurun_out/r06_bwd_probe.txt
export TMPDIR=/tmp
O=/root/repo/gpurun_out; mkdir -p $O
{
echo "# tools/tune.py: C2 kernel times under EPROPNP_TUNE=bwd_mfma=<waves>,<tiles>; base = this tree, oldbwd = the round-5 pair loop"
TUNE_VARIANTS="w4t2:bwd_mfma=4,2|w4t1:bwd_mfma=4,1|w8t2:bwd_mfma=8,2|w8t1:bwd_mfma=8,1|w2t4:bwd_mfma=2,4" timeout 900 python tools/tune.py 2>&1 | cut -c1-200
cd /tmp
for v in base oldbwd; do
  L=""; [ $v = oldbwd ] && L=/root/repo/epro-pnp_amd/lib/variants/oldbwd/libepropnp_hip.so
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_INST_LEVEL_LDS"; do
    i=$((i+1)); rm -rf /tmp/pmc_$v$i
    (EPROPNP_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc_$v$i -o p -- python /root/repo/tools/tune.py --worker 2>&1 | grep -E "rror|ailed" | head -3)
    echo "## $v pass $i"; python /root/repo/tools/pmc_csv_summary.py /tmp/pmc_$v$i 2>&1 | grep -A12 "amis_backward_mfma\|amis_forward_mfma" | head -40
  done
done
} > $O/r06_bwd_probe.txt 2>&1
head -20 $O/r06_bwd_probe.txt
