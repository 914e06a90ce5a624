# Rehearsal of bench.py's multi-rank code path on a one-GPU box: one rank under torch.distributed.run with a real RCCL process
# group, the multi-rank control flow forced on (BENCH_FORCE_DIST=1).  Run twice: the failures it guards against were timing-dependent.
set -u
for port in 29517 29518; do
  BENCH_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/rehearse_$port.json 2> gpurun_out/rehearse_$port.err
  echo "port $port rc=$? stdout_lines=$(wc -l < gpurun_out/rehearse_$port.json)"
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print('  ', d['value'], 'utt/s', d['ms_per_step'], 'ms/step')" gpurun_out/rehearse_$port.json || grep -v "amdgpu.ids\|hostname of the client" gpurun_out/rehearse_$port.err | grep -v "^$" | head -8
done
