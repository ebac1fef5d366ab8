#!/bin/bash
# tools/clock_cap.sh (GPU box, root): the network forward (tools/bench_cnn.py, 200 back-to-back forwards) with the shader clock capped at a series of limits --
# "settle power-vs-skeleton".  Below the clock the chip sustains on its own the time must scale as 1 / cap (the kernels are a fixed number of
# cycles); above it the cap must change nothing (the power limit sets the clock).  Also prints the sampled sclk / socket power during each run.
cd "$GRAFT_REPO_ROOT" || exit 1
rocm-smi --showclocks 2>&1 | grep -i "sclk\|fclk" | head -3
for cap in 0 2100 1900 1700 1500 1300 1100 900; do
  if [ "$cap" == "0" ]; then rocm-smi --resetperfdeterminism > /dev/null 2>&1; else rocm-smi --setperfdeterminism $cap > /tmp/cap.log 2>&1 || { echo "cap $cap: refused"; cat /tmp/cap.log | tail -2; continue; }; fi
  ( for i in 1 2 3 4 5 6; do sleep 0.35; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Average Graphics" | tr '\n' ' '; echo; done ) > /tmp/smi_$cap.log &
  out=$(python tools/bench_cnn.py --steps 600 --warmup 20 "$@" 2>&1 | grep "cnn forward")
  wait
  echo "cap $cap MHz: $out | sampled: $(grep -o 'sclk clock level: [0-9]*: ([0-9]*Mhz)' /tmp/smi_$cap.log | grep -o '[0-9]*Mhz' | sort | uniq -c | tr '\n' ' ') $(grep -o 'Power (W): [0-9.]*' /tmp/smi_$cap.log | sort | uniq -c | tail -2 | tr '\n' ' ')"
done
rocm-smi --resetperfdeterminism > /dev/null 2>&1
