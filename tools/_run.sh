export TMPDIR=/tmp
for v in 308 307; do echo "== VARIANT $v"; VARIANT=$v timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "dgelu"; done
for v in 308 307; do echo "== ATST_TUNE $v"; ATST_TUNE=$v timeout 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | cut -c1-200; done
