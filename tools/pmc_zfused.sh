#!/bin/bash
# rocprofv3 PMC passes over a few 512^3 applies (tools/one_apply.py): SQ activity / wait split and LDS counters per kernel.
# usage (on the GPU box, from the repo root): bash tools/pmc_zfused.sh <outdir under gpurun_out> [n]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
N=${2:-512}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    -d $OUT/p1 --output-format csv -- python3 $R/tools/one_apply.py $N 2 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    -d $OUT/p2 --output-format csv -- python3 $R/tools/one_apply.py $N 2 > $OUT/p2.log 2>&1
python3 - <<EOF
import csv, glob, collections
for p in ("p1", "p2"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("<")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    with open("$OUT/%s_summary.txt" % p, "w") as out:
        for k, d in acc.items():
            if "k_" not in k: continue
            out.write(k + " " + " ".join("%s=%.4g" % kv for kv in sorted(d.items())) + "\n")
    print(open("$OUT/%s_summary.txt" % p).read())
EOF
