#!/usr/bin/env bash
# Collects the round profiles on a GPU box (run through gpurun from the repo root):
#   plain bench line, rocprofv3 kernel stats of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE in
#   separate runs).  Outputs land in gpurun_out/; copy the summaries into profiles/ (see DESIGN.md section 7).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 400 python bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
echo plain done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o p -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/bench_rocprof.json 2> gpurun_out/bench_rocprof.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> gpurun_out/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> gpurun_out/pmc_w.err
echo write done
python tools/summarize_profile.py gpurun_out/prof gpurun_out/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
ls gpurun_out
