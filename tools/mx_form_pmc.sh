#!/bin/bash
# Counters of the matrix path's octave-0 kernel alone (tools/mx_alone.py --octaves 1, 64 frames), both MFMA shapes; separate PMC passes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxform
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so VSLAM_MX=1
for f in 16 32; do
  export VSLAM_MX_FORM=$f
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a$f -o r -- python3 $GRAFT_REPO_ROOT/tools/mx_alone.py --octaves 1 --frames 64 --steps 1 > $OUT/a$f.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/b$f -o r -- python3 $GRAFT_REPO_ROOT/tools/mx_alone.py --octaves 1 --frames 64 --steps 1 > $OUT/b$f.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for form in ("16","32"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for d in ("gpurun_out/mxform/a"+form,"gpurun_out/mxform/b"+form):
        for f in glob.glob(d+"/**/*_counter_collection.csv",recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if "k_pyr_octave_mx" not in k: continue
                acc["k"][r["Counter_Name"]]+=float(r["Counter_Value"])
    v=acc["k"]; wc=max(1,v["SQ_WAVE_CYCLES"]); w=max(1,v["SQ_WAVES"])
    print(f'form {form}: waves {w:9.0f} valu/w {v["SQ_INSTS_VALU"]/w:6.0f} lds/w {v["SQ_INSTS_LDS"]/w:5.0f} mfma/w {v["SQ_INSTS_MFMA"]/w:5.0f} salu/w {v["SQ_INSTS_SALU"]/w:5.0f} | wait_any {v["SQ_WAIT_ANY"]/wc:.2f} wait_inst {v["SQ_WAIT_INST_ANY"]/wc:.2f} wait_lds {v["SQ_WAIT_INST_LDS"]/wc:.2f} act_valu {v["SQ_ACTIVE_INST_VALU"]/wc:.3f} act_lds {v["SQ_ACTIVE_INST_LDS"]/wc:.3f} act_any {v["SQ_ACTIVE_INST_ANY"]/wc:.3f} | mfma_busy/busy {v["SQ_VALU_MFMA_BUSY_CYCLES"]/max(1,v["SQ_BUSY_CYCLES"]):.3f} bank_conf/lds_active {v["SQ_LDS_BANK_CONFLICT"]/max(1,v["SQ_LDS_IDX_ACTIVE"]):.3f} | valu total {v["SQ_INSTS_VALU"]/1e6:.1f}M lds {v["SQ_INSTS_LDS"]/1e6:.1f}M mfma {v["SQ_INSTS_MFMA"]/1e6:.2f}M wave_cycles {wc/1e9:.2f}G busy {v["SQ_BUSY_CYCLES"]/1e6:.1f}M')
PY
