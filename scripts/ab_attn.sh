# alternating A/B of the attention forward kernels on one box: 2 x (streaming, previous), 40 timed steps each
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streaming', d['ms_per_step'], d['value'])"
  SC_ATTN_NO_STREAM=1 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('previous ', d['ms_per_step'], d['value'])"
done
