out=gpurun_out/tb8; mkdir -p $out; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --arch base --workload clip2 --dtype fp8 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-also > /dev/null 2> $out/f.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --arch base --workload clip2 --dtype fp8 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-also > /dev/null 2> $out/w.err
python tools/traffic_from_pmc.py $out/pmc_fetch $out/pmc_write $out/traffic_base_fp8_clip2.json base-fp8-clip2 "$(cat tools/.head)" 12 | tail -12
rm -rf $out/pmc_fetch $out/pmc_write
