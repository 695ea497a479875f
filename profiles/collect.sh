#!/bin/bash
# Round-end measurement on the GPU box: GPU tests (skipped with a second argument "notests"), bench line, rocprofv3
# kernel stats, two PMC passes for HBM traffic and two for the wave/instruction counters (each pass a run of its own).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
T=${1:-r03}
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests_final.txt 2>&1 || { tail -30 $O/gpu_tests_final.txt; exit 1; }
  echo "tests done"
fi
timeout -k 10 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
echo "bench done"
rm -rf $O/prof_$T $O/pmc_${T}_fetch $O/pmc_${T}_write $O/pmc_${T}_sqa $O/pmc_${T}_sqb
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-file-leg > $O/prof_$T.log 2>&1 || exit 1
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${T}_fetch -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_fetch.log 2>&1 || exit 1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${T}_write -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_write.log 2>&1 || exit 1
echo "write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${T}_sqa -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_sqa.log 2>&1 || exit 1
echo "sq a done"
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_${T}_sqb -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_sqb.log 2>&1 || exit 1
echo "sq b done"
