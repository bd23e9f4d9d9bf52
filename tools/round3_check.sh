#!/bin/bash
# GPU check of a round-3 tree: GPU tests, the default bench line, a 4-rank rehearsal of the N > 1 bench path over gloo on
# the one GPU, and a forced-hang rehearsal of the watchdog (must leave with status 3).  Logs under gpurun_out/r3check/.
out=gpurun_out/r3check; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -n 30 $out/pytest.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
cat $out/bench_default.json
export QPSIM_BENCH_BACKEND=gloo QPSIM_BENCH_DEVICE=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 4 --steps 20 --warmup 5 --strong-size 4096 > $out/bench_4rank_gloo.json 2> $out/bench_4rank_gloo.err; echo "4-rank rc=$?"
cat $out/bench_4rank_gloo.json
# watchdog: rank 1 hangs on purpose before its first collective -> line printed with errors, every rank leaves with status 3
QPSIM_BENCH_HANG_RANK=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  bench.py --gpus 2 --steps 5 --warmup 2 --strong-size 4096 --subrecord-timeout 20 > $out/bench_watchdog.json 2> $out/bench_watchdog.err; echo "watchdog rc=$? (expected non-zero)" | tee $out/watchdog.rc
tail -n 5 $out/bench_watchdog.err
