#!/bin/bash
# 2-rank rehearsals of every bench mode on ONE GPU (gloo collectives; timings are meaningless, the flow is what is checked)
export CEL_BENCH_BACKEND=gloo
set -e
python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-sample 0 --legs none | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('weak', d['n_gpus'], d['scaling'], d['value'], d['loglik'])"
python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-sample 0 --scaling strong | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong', d['n_gpus'], d['scaling'], d['value'], d['loglik'])"
python bench.py --gpus 2 --steps 3 --warmup 1 --workload fields8_2048 --cpu-sample 0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fields', d['n_gpus'], d['scaling'], d['value'], d['loglik'])"
python bench.py --gpus 2 --steps 3 --warmup 1 --workload gibbs10k --cpu-sample 0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gibbs weak', d['n_gpus'], d['scaling'], d['value'], d['loglik_trace_tail'])"
python bench.py --gpus 2 --steps 3 --warmup 1 --workload gibbs10k --scaling strong --cpu-sample 0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gibbs strong', d['n_gpus'], d['scaling'], d['value'], d['loglik_trace_tail'], d['sweep_ms'])"
python bench.py --gpus 1 --steps 3 --warmup 1 --workload gibbs10k --cpu-sample 0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gibbs 1 rank, seed as strong?', d['loglik_trace_tail'])"
python bench.py --gpus 2 --steps 3 --warmup 1 --workload gibbs10k --scaling strong --split strips --cpu-sample 0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gibbs strips', d['n_gpus'], d['scaling'], d['value'], d['loglik_trace_tail'], d['sweep_ms'], d['config']['parallelism'][:120])"
