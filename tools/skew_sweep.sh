for mode in 330 332; do for sk in 0 650 1300 2000; do echo "MODE=$mode SKEW=$sk"; ATST_TUNE=$mode,$((100000+sk)) timeout 200 python tools/gemm_bench.py 2>&1 | grep " nt "; done; done
