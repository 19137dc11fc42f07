#!/bin/bash
# Round 6: runs tools/ubench/valu_rate (built here if missing) and one rocprofv3 --pmc pass over a short run of it, writes
# gpurun_out/r06_ubench_valu.txt: the table, the SQ counters per dispatch, and the disassembled loop of the v_fma_f32 variant.
export TMPDIR=/tmp
ROOTDIR=$(pwd); U=$ROOTDIR/tools/ubench
[ -x $U/valu_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $U/valu_rate $U/valu_rate.hip 2>/dev/null
OUT=$ROOTDIR/gpurun_out/r06_ubench_valu.txt
{
echo "== tools/ubench/valu_rate 4000 (timing by s_memtime per wave + HIP events; placement from HW_ID / XCC_ID)"
timeout 300 $U/valu_rate 4000
echo
echo "== SQ counters of a 1000-iteration run (rocprofv3 --pmc, one pass; per dispatch, in launch order = the order of the table above, two launches per line)"
rm -rf $ROOTDIR/gpurun_out/pmc_ubench
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOTDIR/gpurun_out/pmc_ubench -o run -- $U/valu_rate 1000 > $ROOTDIR/gpurun_out/pmc_ubench.log 2>&1)
f=$(find $ROOTDIR/gpurun_out/pmc_ubench -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")))
    d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("dispatch | kernel | wg | grid | SQ_WAVES | SQ_INSTS_VALU/wave | SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (quad-cycles the VALU is busy per wave-instruction; x4 = cycles) | SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES/SQ_WAVES")
for k, c in sorted(d.items()):
    w = c.get("SQ_WAVES", 0) or 1; iv = c.get("SQ_INSTS_VALU", 0) or 1
    print("%4d | %s | %s | %s | %d | %.0f | %.3f | %.0f | %.0f | %.0f" % (k[0], k[1][:14], k[2], k[3], w, iv / w, c.get("SQ_ACTIVE_INST_VALU", 0) / iv, c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0), c.get("SQ_WAVE_CYCLES", 0) / w))
PY
tail -3 $ROOTDIR/gpurun_out/pmc_ubench.log
echo
echo "== ISA of the timed loop, variant 0 (llvm-objdump of the code object in the binary that ran)"
cd /tmp && rm -rf vr_co && mkdir vr_co && cd vr_co && /opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$U/valu_rate >/dev/null 2>&1
/opt/rocm/bin/roc-obj-ls $U/valu_rate 2>/dev/null | head -3
/opt/rocm/bin/roc-obj -d -o /tmp/vr_co $U/valu_rate >/dev/null 2>&1
s=$(find /tmp/vr_co -name "*.s" | head -1)
[ -n "$s" ] && awk '/^[0-9a-f]+ <_Z1kILi0E/,/s_endpgm/' $s | grep -A70 -m1 "s_memtime" | cut -c1-110
} > $OUT 2>&1
find $ROOTDIR/gpurun_out/pmc_ubench -name "*.db" -delete
tail -5 $OUT
