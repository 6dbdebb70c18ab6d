#!/bin/bash
# round 6: the default training step of the ROUND-5 tree (git archive 5ed57b8 under tools/probe/r05_tree, its own library) against the
# current tree, alternating on ONE box (box-to-box spread of the step is 17.1 - 17.9 ms: only a same-box A/B says what a round moved).
cd $GRAFT_REPO_ROOT
one() { (cd $1 && python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-standalone --no-kernel-timing 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['gpu_telemetry']; print('$2  %.3f ms/step  %8.1f samples/s  %s J/step  %.0f W  %.0f MHz' % (d['ms_per_step'], d['value'], ('%.2f' % d['energy_j_per_step']) if d.get('energy_j_per_step') else '    -', t['mean_socket_w'], t['mean_sclk_mhz']))"; }
for i in 1 2 3; do one tools/probe/r05_tree "r05 (5ed57b8)"; one . "r06 (this tree)"; done
