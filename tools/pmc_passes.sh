#!/bin/bash
# tools/pmc_passes.sh <outdir> <kernel-regex> -- <command...>
# One rocprofv3 --pmc pass per counter set (separate passes, --kernel-trace only: the combination
# with other trace domains is refused on this pool), CSV output under <outdir>/<set index>.
out=$1; regex=$2; shift 3
export TMPDIR=/tmp
sets=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_VALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --kernel-include-regex "$regex" --output-format csv -d $out/p$i -- "$@" > $out.p$i.log 2>&1
  i=$((i+1))
done
