cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KW="model=gpu.MODEL_V2, gpu_decode=True"
for L in "" v2ab1 v2ab3 v2ab7; do
  if [ -n "$L" ]; then export AISGPU_LIB=$PWD/tools/ab/$L.so; fi
  echo "== lib ${L:-default}"; LINES_OUT=6 DISTINCT=1 tools/prof_path.sh v2_$L "$KW" 4 256 | grep -E "kv2_engine|kernel "
done > gpurun_out/r05_t5_v2_ablate.txt 2>&1
unset AISGPU_LIB
DISTINCT=1 PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" tools/prof_path.sh v2_pmc "$KW" 2 256 > /dev/null 2>&1
python - <<'PY' >> gpurun_out/r05_t5_v2_ablate.txt 2>&1
import sqlite3, glob
db = glob.glob('/tmp/prof_v2_pmc/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for r in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%kv2_engine%' group by kernel_name, counter_name"):
    print(r[0][:40], r[1], r[2] / r[3])
PY
cat gpurun_out/r05_t5_v2_ablate.txt
