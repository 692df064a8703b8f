for alg in auto rowblock sliced vector noplan; do
timeout 600 python bench.py --gpus 2 --debug-one-gpu --rows 262144 --workload spmv_rmat --steps 3 --warmup 1 --alg $alg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$alg', d['parity']['status'], d['parity']['rows_out_of_bound'], d['config']['plan'].get('alg'), d['config']['plan'].get('n_long_rows'))"
done
