#!/bin/bash
# End-of-round run on the GPU box: GPU tests, smoke, the bench line (default flags = what the driver runs), the two-stream
# leg, the RCCL world-1 run of the graph-captured distributed step, rocprofv3 kernel stats of the bench command and the
# three PMC passes (own runs, program directly after `--`).  Usage (repo root on the box): tools/gpu_validate.sh <tag>
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
# the tree these numbers describe (the box has no .git: the caller passes the commit in SAIS_HEAD)
echo "${SAIS_HEAD:-unknown}" > $O/HEAD
if [ -z "$SAIS_VALIDATE_ONLY_PROFILES" ]; then
if [ -z "$SAIS_VALIDATE_SKIP_TESTS" ]; then python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; tail -3 $O/pytest.log; fi
cp gpurun_out/parity_worst.json $O/parity_worst.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
python bench.py --workload dino > $O/bench_dino.json 2> $O/bench_dino.err; head -c 300 $O/bench_dino.json; echo
python bench.py --two-stream --no-cpu-baseline --sustain-seconds 0 > $O/bench_two_stream.json 2> $O/bench_two_stream.err; head -c 250 $O/bench_two_stream.json; echo
# BASELINE config 5: long-video inference (512 frames, hipGraph extraction + 34 windows x 3 TTA + attention export)
python bench.py --workload extract --steps 20 --warmup 3 > $O/bench_extract.json 2> $O/bench_extract.err; head -c 300 $O/bench_extract.json; echo
# the distributed branch of bench.py over RCCL with a world of one: the step INCLUDING its all-reduces is one hipGraph
SAIS_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --sustain-seconds 0 > $O/force_dist_world1.log 2>&1; tail -c 2500 $O/force_dist_world1.log | head -c 300; echo
SAIS_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --workload dino --gpus 1 --steps 10 --warmup 3 > $O/force_dist_world1_dino.log 2>&1; tail -c 1200 $O/force_dist_world1_dino.log | head -c 400; echo
fi   # SAIS_VALIDATE_ONLY_PROFILES
# --no-variants below: the config 4 / 5 legs of the default run are child processes and would be profiled into the same files
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --parity-clips 0 --no-variants > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_extract -- python3 $R/bench.py --workload extract --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_extract.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dino -- python3 $R/bench.py --workload dino --steps 10 --warmup 3 --no-cpu-baseline > $O/stats_dino.log 2>&1
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --sustain-seconds 0 --parity-clips 0 --no-variants"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
          SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
# vector-L1 / L2 request counters (three more passes): how many bytes a kernel moves through its CUs' L1s (LABNOTES R5.5, R6.1)
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $O/tcp1 -- $B > $O/tcp1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $O/tcp2 -- $B > $O/tcp2.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum --kernel-trace --output-format csv -d $O/tcp3 -- $B > $O/tcp3.log 2>&1
# the counter reports are made HERE (the raw per-dispatch CSVs are too large to travel: gpurun merges <= 64 MiB) and copied next to
# the stats; profiles/ of the build container receives them from gpurun_out/<tag>/
cd $R
python tools/pmc_report.py $O/pmc_sq $O/pmc_fetch $O/pmc_write > $O/pmc_report.log 2>&1; tail -30 $O/pmc_report.log
python tools/pmc_tcp_report.py $O/tcp1 $O/tcp2 $O/tcp3 > $O/pmc_tcp_report.log 2>&1; grep -E "gemm_tn_grouped|gelu_grad|attn_bwd" $O/pmc_tcp_report.log
cp profiles/pmc_mfma.json profiles/pmc_traffic.json profiles/r06_pmc_tcp.json $O/ 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
