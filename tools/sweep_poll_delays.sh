#!/bin/bash
# Sweep the first-poll delays of the persistent GRU scans (PBSED_GRU_POLL_DELAYS="fwd,fwd_gate,bwd,bwd_gate", 64-clock units)
# with the headline bench: prints clips/s and the two scan durations per setting.  usage: tools/sweep_poll_delays.sh
for f in ${FWD:-24 25 26 27 28}; do for b in ${BWD:-16 18 20}; do
  PBSED_GRU_POLL_DELAYS="$f,6,$b,0" python bench.py --config ${CFG:-c2} --no-cpu-baseline --headline-only --steps 20 --warmup 5 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['roofline_gru']; print('fwd_delay $f bwd_delay $b ->', d['value'], g['forward_scan']['ms_per_step'], g['bptt_scan']['ms_per_step'])"
done; done
