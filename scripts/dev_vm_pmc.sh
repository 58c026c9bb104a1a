# developer aid: PMC counters of the step kernels behind scripts/dev_time_vm.py (native / interpreted / jit)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/vm_pmc
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  d=$R/gpurun_out/vm_pmc/$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -o run -- python3 $R/scripts/dev_time_vm.py > $d.log 2>&1
  echo "$grp rc=$?"
done
python3 - <<'PY'
import csv, glob, os, statistics
R=os.environ["GRAFT_REPO_ROOT"]
res={}
for f in glob.glob(R+"/gpurun_out/vm_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "lm_step_kernel" not in k: continue
        short="VM-lds" if "ModelVM<3, true>" in k or "ModelVMILi3ELb1" in k else ("VM" if "ModelVM" in k else ("JIT" if "ModelJit" in k else "native"))
        jac = k.split("ModelVM")[-1] if False else ""
        key=short+" "+k.split(",")[1 if short=="native" or short=="JIT" else 2].strip()[:3]
        res.setdefault(key,{}).setdefault(row["Counter_Name"],[]).append(float(row["Counter_Value"]))
for k,v in sorted(res.items()):
    print(k, {c: round(statistics.median(x)) for c,x in sorted(v.items())}, "n=%d" % len(next(iter(v.values()))))
PY
