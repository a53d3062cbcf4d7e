#!/bin/bash
# One parameterised GPU-box script (round 3 on; replaces the one-shot gpu_r2?.sh of round 2):
#   gpurun --timeout T -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
# Steps run in order, each with its own timeout, everything logged to gpurun_out/<tag>.log:
#   env:K=V / unset:K     set / clear a library knob (scripts/README.md) for the steps that follow
#   tests[:<-k expr>]     pytest -m gpu (-x), optionally restricted
#   smoke                 __graft_entry__.smoke()
#   bench[:N:L[:steps]]   bench.py --no-cpu-baseline on the synthetic family -> gpurun_out/<tag>_bench_<N>x<L>[_<label>].json
#   benchcpu              the driver's default command (CPU baseline + parity self-check) -> gpurun_out/<tag>_bench_default.json
#   rdrp[:n]              bench.py --fasta tests/golden/rdrp_first1000.fa.gz --n n (default 1000), one step
#   stats[:N:L]           rocprofv3 --kernel-trace --stats of one bench step -> gpurun_out/<tag>_kernel_stats_<N>x<L>.csv
#   pmc[:N:L]             separate --pmc passes (FETCH_SIZE, WRITE_SIZE, two SQ sets) of one bench step + scripts/pmc_summary.py
#   rdrpstats / rdrppmc[:n] the same two on the rdrp input
#   label:<text>          suffix for the output files of the following bench steps (A/B runs)
#   sh:<command>          anything else (quoted as one argument)
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out; mkdir -p $OUT; LOG=$OUT/$TAG.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LABEL=""
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-12}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
RDRP=$R/tests/golden/rdrp_first1000.fa.gz
prof() { # prof <dir> <rocprof args...> -- <bench args...>
	local d=$1; shift; rm -rf $OUT/$d
	local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
	( cd /tmp && TAILN=3 step timeout 600 rocprofv3 "${pa[@]}" --output-format csv -d $OUT/$d -o r -- python -u $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-real-data )
}
for s in "$@"; do
	IFS=: read -r name a1 a2 a3 <<< "$s"
	case $name in
	env) export "${s#env:}"; echo "=== env ${s#env:}" | tee -a $LOG;;
	unset) unset "$a1"; echo "=== unset $a1" | tee -a $LOG;;
	label) LABEL="_$a1";;
	tests) if [ -n "${a1:-}" ]; then TAILN=15 step timeout 1500 python -u -m pytest tests -m gpu -q -x -k "${s#tests:}"; else TAILN=15 step timeout 1500 python -u -m pytest tests -m gpu -q -x; fi;;
	smoke) step timeout 120 python -u -c "import __graft_entry__ as g; g.smoke()";;
	bench) N=${a1:-1000}; L=${a2:-400}; K=${a3:-2}
		step timeout 900 python -u bench.py --n $N --len $L --steps $K --warmup 1 --no-cpu-baseline --no-real-data
		grep '^{"metric"' $LOG | tail -1 > $OUT/${TAG}_bench_${N}x${L}${LABEL}.json;;
	benchcpu) step timeout 900 python -u bench.py
		grep '^{"metric"' $LOG | tail -1 > $OUT/${TAG}_bench_default${LABEL}.json;;
	rdrp) N=${a1:-1000}
		step timeout 900 python -u bench.py --fasta $RDRP --n $N --steps 1 --warmup 1 --no-cpu-baseline
		grep '^{"metric"' $LOG | tail -1 > $OUT/${TAG}_bench_rdrp${N}${LABEL}.json;;
	stats) N=${a1:-1000}; L=${a2:-400}
		prof ${TAG}_prof_stats --kernel-trace --stats -- --n $N --len $L
		for f in $(find $OUT/${TAG}_prof_stats -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${TAG}_kernel_stats_${N}x${L}.csv; head -14 $f | cut -c1-220 | tee -a $LOG; done
		find $OUT/${TAG}_prof_stats -name "*kernel_trace.csv" -size +20M -delete;;
	rdrpstats) N=${a1:-1000}
		prof ${TAG}_prof_stats_rdrp --kernel-trace --stats -- --fasta $RDRP --n $N
		for f in $(find $OUT/${TAG}_prof_stats_rdrp -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${TAG}_kernel_stats_rdrp${N}.csv; head -14 $f | cut -c1-220 | tee -a $LOG; done
		find $OUT/${TAG}_prof_stats_rdrp -name "*kernel_trace.csv" -size +20M -delete;;
	pmc|rdrppmc)
		if [ $name = pmc ]; then N=${a1:-1000}; L=${a2:-400}; BA=(--n $N --len $L); SUF=${N}x${L}; else N=${a1:-1000}; L=0; BA=(--fasta $RDRP --n $N); SUF=rdrp$N; fi
		prof ${TAG}_pmc_fetch_$SUF --pmc FETCH_SIZE -- "${BA[@]}"
		prof ${TAG}_pmc_write_$SUF --pmc WRITE_SIZE -- "${BA[@]}"
		prof ${TAG}_pmc_sq_$SUF --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- "${BA[@]}"
		prof ${TAG}_pmc_sq2_$SUF --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -- "${BA[@]}"
		for k in fetch write sq sq2; do for f in $(find $OUT/${TAG}_pmc_${k}_$SUF -name "*counter_collection.csv" | head -1); do cp $f $OUT/${TAG}_pmc_${k}_${SUF}_counter_collection.csv; done; done
		prof ${TAG}_pmc_grbm_$SUF --pmc GRBM_GUI_ACTIVE -- "${BA[@]}"
		for f in $(find $OUT/${TAG}_pmc_grbm_$SUF -name "*counter_collection.csv" | head -1); do cp $f $OUT/${TAG}_pmc_grbm_${SUF}_counter_collection.csv; done
		# (run `stats` / `rdrpstats` BEFORE this step: its kernel_stats csv gives the launch durations the clock is derived with)
		step env PMC_STATS=$OUT/${TAG}_kernel_stats_$SUF.csv PMC_GRBM=$OUT/${TAG}_pmc_grbm_$SUF PMC_SOURCE="profiles/${TAG}_pmc_*_${SUF}_counter_collection.csv, profiles/${TAG}_kernel_stats_$SUF.csv" python scripts/pmc_summary.py $N $L $OUT/${TAG}_pmc_fetch_$SUF $OUT/${TAG}_pmc_write_$SUF $OUT/${TAG}_pmc_traffic_$SUF.json $OUT/${TAG}_pmc_sq_$SUF $OUT/${TAG}_pmc_sq2_$SUF;;
	sh) step timeout 1500 bash -c "${s#sh:}";;
	*) echo "gpu.sh: unknown step $s" | tee -a $LOG;;
	esac
done
echo "=== done (t=$SECONDS)" | tee -a $LOG
