#!/bin/bash
# Round-end measurement on the GPU box: GPU tests, bench line, rocprofv3 kernel stats, two PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
T=${1:-r02}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests_final.txt 2>&1 || { tail -30 $O/gpu_tests_final.txt; exit 1; }
echo "tests done"
timeout -k 10 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
echo "bench done"
rm -rf $O/prof_$T $O/pmc_${T}_fetch $O/pmc_${T}_write
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-file-leg > $O/prof_$T.log 2>&1 || exit 1
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${T}_fetch -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_fetch.log 2>&1 || exit 1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${T}_write -o $T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > $O/pmc_${T}_write.log 2>&1 || exit 1
echo "write done"
