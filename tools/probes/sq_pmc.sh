#!/bin/bash
# SQ counters of the recurrent kernels (one rocprofv3 --pmc pass per group); run on the GPU box.
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out/sq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
short="python $repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $out/avail.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o g -- $short > $out/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/g*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "rec_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
