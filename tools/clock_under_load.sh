#!/bin/bash
# GPU box: shader clock, power and temperature sampled by rocm-smi twice a second, idle and then while the DeiT-S bench runs
# (the forward against the board's power limit: profiles/README.md, round 6, "the clock follows the activity")
smp() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power|junction|hotspot" | tr -s ' ' | tr '\n' '|'; echo; }
echo "== idle"; for i in 1 2 3; do smp; sleep 0.5; done
python bench.py --no-cpu-baseline --measure-traffic 0 --profile-steps 0 --box-probe 0 --steps 1500 --reps 3 --min-seconds 1 --streams 1 --graph 0 > /tmp/clk_bench.json 2>/dev/null &
pid=$!
echo "== while bench.py runs (imports and weights first: ~8 s, then 3 x 1500 forwards)"; while kill -0 $pid 2>/dev/null; do smp; sleep 0.5; done
tail -1 /tmp/clk_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"
