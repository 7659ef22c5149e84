#!/bin/bash
# round 4, call J: where k_solve's waves wait — instruction fetch, scalar / memory issue cycles, FIFO back-pressure
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04j; mkdir -p $o
S="python3 bench.py --config 1 --tsteps 1920 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify"
rocprofv3 --kernel-trace --output-format csv -d $o/p1 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD -- $S > /dev/null 2> $o/p1.err; echo p1
rocprofv3 --kernel-trace --output-format csv -d $o/p2 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU -- $S > /dev/null 2> $o/p2.err; echo p2
rocprofv3 --kernel-trace --output-format csv -d $o/p3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES -- $S > /dev/null 2> $o/p3.err; echo p3
rocprofv3 --kernel-trace --output-format csv -d $o/p4 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES GRBM_GUI_ACTIVE -- $S > /dev/null 2> $o/p4.err; echo p4
python3 - <<P
import csv, glob, collections
for p in ("p1","p2","p3","p4"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_solve<" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(p, k, "%.4e" % (sum(v)/len(v)), len(v))
P
