#!/bin/bash
# round 6, session 2, call 9: single recurrent minibatch in buffer order (generator advanced, tables and input image built once
# per update): recurrent goldens, sharded runs, the full-size SMAC / HATRPO-GRU checks, A/B on the SMAC shapes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider -k "recurrent or gru or rnn or smac3s5z or sharded or mappo or rollout_loop or trunk or dropin or run_with_eval or generator_api" ) > $O/t9.txt 2>&1
tail -6 $O/t9.txt
for cfg in smac3s5z smac3s5z_n4096; do for f in 1 0 1 0; do
HARL_RNN_ORDERED=$f timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-cols 0 --instr-steps 0 --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg ordered=$f', round(d['ms_per_step'],3))"
done; done
HARL_RNN_ORDERED=1 timeout 600 python bench.py --config hatrpo_gru128 --steps 3 --warmup 1 --cpu-cols 0 --instr-steps 0 --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hatrpo_gru128', round(d['ms_per_step'],3))"
