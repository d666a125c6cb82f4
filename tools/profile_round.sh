#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + three PMC passes of the SAME bench command (GPU box).
#   GIT_HEAD=<sha> tools/profile_round.sh r02 [precision]  -> profiles/<tag>_kernel_stats.{txt,json}, profiles/<tag>_pmc[_prec].json
# (copied to gpurun_out/profiles/ so they come back from the GPU box; GIT_HEAD is recorded as provenance)
# Counters are collected in their own runs (--kernel-trace + --pmc only), as the pool requires.
set +e
TAG=${1:-r01}
PREC=${2:-bf16x6}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-profile --no-modes --no-workloads --repeats 1 --precision $PREC"
O=$R/gpurun_out
SUF=""; [ "$PREC" != "bf16x6" ] && SUF="_$PREC"
rm -rf $O/prof_$TAG$SUF $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG$SUF -o bench -- $B > $O/prof_$TAG$SUF.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $B > $O/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $B > $O/pmc2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- $B > $O/pmc3.log 2>&1
cd $R
DB=$(ls $O/prof_$TAG$SUF/*/*.db $O/prof_$TAG$SUF/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py "$DB" profiles/${TAG}_kernel_stats$SUF
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq profiles/${TAG}_pmc$SUF "$PREC" "${GIT_HEAD:-unknown}" "${B#python3 $R/}"
mkdir -p $O/profiles && cp profiles/${TAG}_* $O/profiles/
tail -1 $O/prof_$TAG$SUF.log
