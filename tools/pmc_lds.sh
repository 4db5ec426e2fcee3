#!/bin/bash
# LDS bank-conflict share per kernel over a short bench run (GPU box)
export TMPDIR=/tmp
out=gpurun_out/pmclds; rm -rf $out; mkdir -p $out
timeout 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/a -o a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-also ${WL:+--workload $WL} $ARGS > /dev/null 2>&1
python - "$out" <<'PY'
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:48]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:18]
print(f"{'kernel':48s} {'conflict/LDSactive':>18s} {'wait/wave':>10s} {'valu/wave':>10s} {'mfma-busy/wave-cycle':>20s}")
for n, v in rows:
    wc = v["SQ_WAVE_CYCLES"] or 1
    print(f"{n:48s} {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):18.3f} {v['SQ_WAIT_INST_ANY'] / wc:10.3f} {v['SQ_ACTIVE_INST_VALU'] / wc:10.3f} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc):20.3f}")
PY
rm -rf $out
