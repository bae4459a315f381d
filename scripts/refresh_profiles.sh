#!/bin/bash
# re-measure every committed line under profiles/ with the current build (run on the GPU box; outputs under gpurun_out/refresh/).
# usage: bash scripts/refresh_profiles.sh <round tag, e.g. round3>      (every number quoted in DESIGN.md / README.md comes from a file this writes)
# other scripts kept here: stress_matrix_types.py, hqrrp_tall_check.py, trsm_bench.py, saso_time.py, linops_time.py
#   qrcp_wide_parts.py + lu_only.py (per-part timing of BQRRP's qrcp_wide step / the LU alone: DESIGN 4.10), ld_ab.py (leading-dimension A/B at C3's shape: DESIGN 4.10)
R=$GRAFT_REPO_ROOT
TAG=${1:-round6}
export PYTHONPATH=$R
O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
# counter evidence FIRST (three rocprofv3 --pmc passes per kernel; cooperative launches are refused by rocprofv3 --pmc: the persistent Jacobi goes
# through its ordinary-launch route, option jacobi_persist = 3): the bench lines below quote a counter file only when its launch time is within
# 5 % of the live one, so they must see the files of THIS build
( timeout 1800 python scripts/pmc_all.py $TAG > $O/pmc.log 2>&1; cp gpurun_out/pmc/${TAG}_pmc_*.json $O/ 2>/dev/null; cp gpurun_out/pmc/${TAG}_pmc_*.json $R/profiles/ 2>/dev/null )
timeout 300 python bench.py < /dev/null > $O/${TAG}_bench_line.json 2> $O/bench.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $O/${TAG}_c3_cqrrpt_line.json 2> $O/c3.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 3 --opt saso_mode=0 < /dev/null > $O/${TAG}_c3_cqrrpt_affine_saso_line.json 2>> $O/c3.err
timeout 300 python scripts/bench_other.py bqrrp64 --steps 3 < /dev/null > $O/${TAG}_bqrrp_f64_16k_line.json 2> $O/b64.err
timeout 300 python scripts/bench_other.py bqrrp_full --steps 2 < /dev/null > $O/${TAG}_c4_bqrrp_f32_65536_line.json 2> $O/c4.err
timeout 300 python scripts/bench_other.py bqrrp_full --steps 1 --triple default < /dev/null > $O/${TAG}_c4_bqrrp_f32_65536_default_triple_line.json 2>> $O/c4.err
timeout 300 python scripts/bench_other.py abrik --steps 2 < /dev/null > $O/${TAG}_c5_abrik_line.json 2> $O/c5.err
timeout 400 python scripts/bench_other.py rsvd_p2 --steps 2 < /dev/null > $O/${TAG}_c2_p2_planted_line.json 2> $O/c2p2.err
timeout 300 python bench.py --m 25000 --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/${TAG}_rank_of_8_line.json 2> $O/r8.err
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...
    local name=$1; shift
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- "$@" < /dev/null > $O/${TAG}_${name}_line_profiled.json 2> $O/prof_$name.err
    local f=$(find $O/prof_$name -name '*kernel_stats.csv' 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $O/${TAG}_${name}_kernel_stats.csv; fi
    rm -rf $O/prof_$name
}
prof bench python $R/bench.py --no-cpu-baseline
prof c3_cqrrpt python $R/scripts/bench_other.py cqrrpt --steps 3
prof rank_of_8 python $R/bench.py --m 25000 --steps 5 --warmup 3 --no-cpu-baseline
prof c4_bqrrp_f32_65536 python $R/scripts/bq_prof.py 65536 2048 f32
# kernel timeline of one step of the 1/8 row shard (start, duration, gap before each kernel)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python $R/bench.py --m 25000 --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/timeline.py $O/tl 5 > $O/${TAG}_rank_of_8_timeline.txt 2>&1; rm -rf $O/tl
# the DVFS map behind the clock holders and the look-ahead A/B of BQRRP
timeout 300 python $R/scripts/dvfs_probe.py > /dev/null 2> $O/${TAG}_dvfs_probe.txt
timeout 600 python $R/scripts/bqrrp_lookahead_ab.py 65536 2048 f32 2 > $O/${TAG}_c4_lookahead_ab.txt 2>&1
# second half of round 4: C3 with and without the split QRCP + the timeline of one call, the sparse product A/B,
# the cooperative Householder kernels before / after the address-space fix need the previous library and are not re-run here (profiles/round4_qr_addrspace_ab.txt)
( cd $R && bash scripts/c3_split_evidence.sh $TAG > $O/c3split.log 2>&1; cp gpurun_out/c3split/${TAG}_c3_* $O/ 2>/dev/null )
# the real sharded drivers as 8 ranks on this one device: per-rank time = 1/8 of the rows + every replicated stage
cd $R
for w in rsvd cqrrpt bqrrp abrik; do timeout 600 python scripts/ranks_on_one_device.py $w --check --steps 2 > $O/${TAG}_ranks8_on_one_device_$w.json 2> $O/ranks8_$w.err; done
for j in $O/${TAG}_*line.json; do echo "$(basename $j): $(cut -c1-240 $j)"; done
