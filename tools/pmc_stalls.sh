#!/bin/bash
# Stall counters of the walk's kernels and the rasteriser: TLB (UTCL1), L1 <-> L2 latency, workgroup-launch stalls.  One rocprofv3 --pmc pass per
# small set (a set the hardware cannot collect at once makes rocprofv3 abort and hang: every pass runs under its own timeout), no trace domains.
# tools/pmc_stalls.sh <tag> [bench.py flags]
TAG=${1:-stalls}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmc_$TAG
BENCH="python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency $*"
export TMPDIR=/tmp
mkdir -p $OUT
pass() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/$name.log 2>&1 && echo "$name ok" || echo "$name FAILED ($?)"; }
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass pend TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pass spi1 SPI_RA_RES_STALL_CSN SPI_RA_LDS_CU_FULL_CSN
pass spi2 SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN
pass spi3 SPI_RA_BAR_CU_FULL_CSN SPI_CSN_BUSY
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT "stall counters" | grep "dg_fe_columns\|dg_fe_scatter\|dg_raster_tiles\|dg_fe_scan\|dg_fe_gaps"
