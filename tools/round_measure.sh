#!/bin/bash
# Round measurement pass on the GPU box: bench lines, rocprofv3 kernel stats of the default bench command, PMC traffic passes.
# usage (inside gpurun): bash tools/round_measure.sh <tag>
# The rocprofv3 passes run `bench.py --no-also`: the kernel stats are the headline workload's alone (round 5's mixed in the five `also` workloads).
tag=${1:-r06}
head=${2:-$(cat tools/.head 2>/dev/null || echo unknown)}      # the GPU box has no .git: pass HEAD as $2 (or write tools/.head before the call)
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py > $out/bench_clip6.json 2> $out/bench_clip6.err
timeout 300 python bench.py --workload clip2 --no-cpu-baseline > $out/bench_clip2.json 2> $out/bench_clip2.err
timeout 300 python bench.py --workload frame --no-cpu-baseline > $out/bench_frame.json 2> $out/bench_frame.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o clip6 -- python3 bench.py --no-also > $out/bench_clip6_under_rocprof.json 2> $out/prof.err
# the same command with the second stream off (ATST_OVERLAP_LT=0): per-kernel durations without the time a launch shares the chip with the other chain
ATST_OVERLAP_LT=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_excl -o clip6_exclusive -- python3 bench.py --no-cpu-baseline --no-also > $out/bench_clip6_exclusive_under_rocprof.json 2> $out/prof_excl.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> $out/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> $out/pmc_write.err
python tools/traffic_from_pmc.py $out/pmc_fetch $out/pmc_write $out/traffic_clip6.json clip6 "$head" 3
# round 6: the all-e4m3 step at d = 384 (bench line + whole-step PMC traffic next to the bf16 one), ATST-Frame base
timeout 300 python bench.py --dtype fp8 --no-cpu-baseline --no-also > $out/bench_small_fp8_clip6.json 2> $out/bench_small_fp8.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch8 -o f -- python3 bench.py --dtype fp8 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-also > /dev/null 2> $out/pmc_fetch8.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write8 -o w -- python3 bench.py --dtype fp8 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-also > /dev/null 2> $out/pmc_write8.err
python tools/traffic_from_pmc.py $out/pmc_fetch8 $out/pmc_write8 $out/traffic_small_fp8_clip6.json clip6-fp8 "$head" 5
timeout 300 python bench.py --arch base --workload frame --no-cpu-baseline --no-also > $out/bench_base_frame.json 2> $out/bench_base_frame.err
timeout 300 python bench.py --arch base --workload frame --dtype fp8 --no-cpu-baseline --no-also > $out/bench_base_fp8_frame.json 2>> $out/bench_base_frame.err
# extra data points: ATST-base bf16 / fp8 / fp8 at the configs[4] input geometry ; the fp32 parity mode next to bf16 at the same batch
timeout 300 python bench.py --arch base --workload clip2 --no-cpu-baseline > $out/bench_base_clip2.json 2> $out/bench_base.err
timeout 300 python bench.py --arch base --workload clip2 --dtype fp8 --no-cpu-baseline > $out/bench_base_fp8_clip2.json 2>> $out/bench_base.err
timeout 300 python bench.py --arch base --workload clip2 --dtype fp8 --hires --no-cpu-baseline > $out/bench_base_fp8_hires_clip2.json 2>> $out/bench_base.err
timeout 600 python bench.py --precise --workload clip2 --batch 64 --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_precise_clip2_b64.json 2> $out/bench_precise.err
timeout 300 python bench.py --workload clip2 --batch 64 --steps 40 --no-cpu-baseline --no-profile > $out/bench_bf16_clip2_b64.json 2>> $out/bench_precise.err
find $out -name "*_kernel_stats.csv" | head; cat $out/bench_clip6.json | cut -c1-400
# the raw per-dispatch counter CSVs are large: keep only the summary json in the merge-back
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_fetch8 $out/pmc_write8
find $out/prof $out/prof_excl -type f ! -name "*_kernel_stats.csv" -delete
