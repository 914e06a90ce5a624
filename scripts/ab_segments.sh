# one-rank rehearsal of the multi-rank step: backward replayed as five hipGraphs with bucket all-reduces in between, vs one graph and all
# buckets afterwards (run through gpurun from the repo root).  On one rank an all-reduce has no peers: this measures the OVERHEAD of the pieces.
set -u
mkdir -p gpurun_out/seg
port=29560
for rep in 1 2; do
  for mode in 0 1; do
    port=$((port + 1))
    BENCH_NO_SEGMENTS=$mode BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 1 --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/seg/m${mode}_$rep.json 2> gpurun_out/seg/m${mode}_$rep.err
    python -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print('no_segments=$mode rep $rep', d['ms_per_step'], 'ms/step')" gpurun_out/seg/m${mode}_$rep.json || tail -5 gpurun_out/seg/m${mode}_$rep.err
  done
done
python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/seg/plain.json 2> gpurun_out/seg/plain.err
python -c "import json; d=json.load(open('gpurun_out/seg/plain.json')); print('plain single-process', d['ms_per_step'])"
