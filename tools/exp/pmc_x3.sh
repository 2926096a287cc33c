#!/bin/bash
# PMC passes over tools/exp/x3_vit_time.py; prints per-kernel means.   usage: pmc_x3.sh <outdir-tag> <kernel-substring>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; KN=$2
mkdir -p $R/gpurun_out/r05/$TAG; cd /tmp; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/r05/$TAG/$n -o out --output-format csv -- python $R/tools/exp/x3_vit_time.py 30 1 > /dev/null 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r05/$TAG/*/*counter_collection.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "$KN" in r["Kernel_Name"]]
    ids=sorted(set(int(r["Dispatch_Id"]) for r in rows))
    for r in rows:
        k=ids.index(int(r["Dispatch_Id"]))//3
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d={n:sum(v)/len(v) for n,v in acc[k].items()}
    print("shape",k," ".join(f"{n}={v:.4g}" for n,v in d.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        print("   matrix-pipe busy %.3f; LDS conflict share %.3f; wave wait share %.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(d["GRBM_GUI_ACTIVE"]/8), d.get("SQ_LDS_BANK_CONFLICT",0)/max(d.get("SQ_LDS_IDX_ACTIVE",1),1), d.get("SQ_WAIT_INST_ANY",0)/max(d.get("SQ_WAVE_CYCLES",1),1)))
PY
