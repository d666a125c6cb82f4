#!/bin/bash
# rocprofv3 evidence for profiles/ of ANY python command of this repo (GPU box): a kernel trace + three PMC passes.
#   GIT_HEAD=<sha> tools/profile_cmd.sh <tag> <name> <precision-label> <script.py> [args...]
#     -> profiles/<tag>_kernel_stats_<name>.{txt,json}, profiles/<tag>_pmc_<name>.json   (copied to gpurun_out/profiles/ too)
# e.g. tools/profile_cmd.sh r03 dit f16 tools/workloads.py --only dit
# Counters are collected in their own runs (--kernel-trace + --pmc only), as the pool requires; the program itself follows
# `--` (no shell / env hop between the profiler and python).
set -e
TAG=$1; NAME=$2; PREC=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
D=$O/prof_${TAG}_$NAME
rm -rf $D ${D}_fetch ${D}_write ${D}_sq
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D -o run -- python3 $SCRIPT "$@" > $D.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${D}_fetch -o p -- python3 $SCRIPT "$@" > ${D}_pmc1.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${D}_write -o p -- python3 $SCRIPT "$@" > ${D}_pmc2.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d ${D}_sq -o p -- python3 $SCRIPT "$@" > ${D}_pmc3.log 2>&1
cd $R
DB=$(ls $D/*/*.db $D/*.db 2>/dev/null | head -1)
python3 tools/rocprof_summary.py "$DB" profiles/${TAG}_kernel_stats_$NAME > /dev/null
python3 tools/pmc_summary.py $(dirname $(ls ${D}_fetch/*/p_counter_collection.csv ${D}_fetch/p_counter_collection.csv 2>/dev/null | head -1)) \
    $(dirname $(ls ${D}_write/*/p_counter_collection.csv ${D}_write/p_counter_collection.csv 2>/dev/null | head -1)) \
    $(dirname $(ls ${D}_sq/*/p_counter_collection.csv ${D}_sq/p_counter_collection.csv 2>/dev/null | head -1)) \
    profiles/${TAG}_pmc_$NAME "$PREC" "${GIT_HEAD:-unknown}" "${SCRIPT#$R/} $*" > /dev/null
mkdir -p $O/profiles && cp profiles/${TAG}_kernel_stats_$NAME.* profiles/${TAG}_pmc_$NAME.json $O/profiles/
head -12 profiles/${TAG}_kernel_stats_$NAME.txt
