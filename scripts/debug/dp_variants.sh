#!/bin/bash
# the one-rank data-parallel step in its forms (collectives forced / not issued; flush graph / inline; captured hook), then the one-graph step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/dp
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 1 --force-dist --no-cpu-baseline --no-secondary --no-roofline "$@" > $R/gpurun_out/dp/$name.log 2>&1
  echo "$name $(grep -h '^{' $R/gpurun_out/dp/$name.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
}
run graph_forced X=1 --
run inline_forced MISEG_SPLIT_FLUSH=inline --
run graph_nocoll MISEG_FORCE_COLLECTIVE=0 --
run inline_nocoll MISEG_FORCE_COLLECTIVE=0 MISEG_SPLIT_FLUSH=inline --
run hook_forced X=1 -- --captured-collective
run hook_nocoll MISEG_FORCE_COLLECTIVE=0 -- --captured-collective
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline > $R/gpurun_out/dp/one.log 2>&1; echo "one $(grep -h '^{' $R/gpurun_out/dp/one.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
