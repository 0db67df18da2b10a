#!/bin/bash
# tools/profile_round.sh <tag>: everything the round's profiles/ summaries come from, in one gpurun call:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r04'
# then  python tools/summarize_round.py gpurun_out/prof_r04 r04
#  1. the default `python bench.py` run, plain (the bench line) and under rocprofv3 --kernel-trace --stats
#  2. PMC passes (one counter set per pass, --kernel-trace only) of the two correlator builds alone
#     (tools/native/corrbench, no torch in the process) and of the timing-recovery kernel in a short chain run
#  3. FETCH_SIZE / WRITE_SIZE passes of a short whole-flowgraph run (bytes per kernel and step)
tag=${1:-r06}
out=gpurun_out/prof_$tag
mkdir -p $out
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.log 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/bench_profiled.log 2> $out/bench_profiled.err
# 2. correlators alone
bash tools/pmc_passes.sh $out/corr896 k_corr4f -- tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 5
bash tools/pmc_passes.sh $out/corr112 k_corr2d -- tools/native/corrbench gr-ais_amd/lib/libaisx.so --N 112 --iters 5
python tools/pmc_table.py $out/corr896 k_corr4f > $out/corr896_pmc_table.json
python tools/pmc_table.py $out/corr112 k_corr2d > $out/corr112_pmc_table.json
# the timing-recovery kernel inside the chain (SQ counters)
for s in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $s --kernel-include-regex "k_msk<" --output-format csv -d $out/msk/p$i -- python bench.py --steps 4 --warmup 2 --single-chain --no-cpu-baseline --parity-channels 0 > $out/msk.p$i.log 2>&1
done
python tools/pmc_table.py $out/msk "k_msk<" > $out/msk_pmc_table.json
# the front-end kernels with the chip to themselves
i=0
for s in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  rocprofv3 --kernel-trace --pmc $s --kernel-include-regex "k_agcw|k_fs_est" --output-format csv -d $out/front/p$i -- python tools/front_alone.py > $out/front.p$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_table.py $out/front "k_agcw" > $out/agcw_pmc_table.json
python tools/pmc_table.py $out/front "k_fs_est" > $out/est_pmc_table.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/front_stats -- python tools/front_alone.py > $out/front_stats.log 2>&1
./tools/ubench/hbm_ceiling json > $out/hbm_ceiling.json 2> $out/hbm_ceiling.err
# 3. bytes per kernel of the whole step
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/chain_$c -- python bench.py --steps 3 --warmup 2 --single-chain --no-cpu-baseline --parity-channels 0 > $out/chain_$c.log 2>&1
done
# 4. round 6: config 4's per-GPU shape under the kernel trace; the correlator's energy split (CE_DBG builds: timing / power
#    only); the front-end claim sweep
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4stats -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $out/c4_profiled.log 2> $out/c4_profiled.err
python tools/corr_energy.py $out/energy > $out/energy.log 2>&1
timeout 900 python tools/claim_sweep.py --claims 0,16,24,32,40,48,56,63,72 --steps 20 --reps 3 > $out/claim_sweep.jsonl 2> $out/claim_sweep.err
tail -c 400 $out/bench_default.log
