#!/bin/bash
# Phase-aware tile sweep (round 5): the text stream's GEMM sites in the CO-ATTENTION phase only (phase 1: the visual stream runs beside them
# and is the longer chain there, profiles/r5_step_sensitivity.txt), bench.py --site-policy site:kind:phase:cfg:split_k.  ms per step.
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --no-h2d-leg --sustained-s 0"
run() { echo -n "$1: "; $B ${2:+--site-policy $2} 2>/dev/null | grep '^{' | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
T1w4="t.ffn_up:fwd:1:4:1,t.ffn_down:dgrad:1:4:1,t.qkv:fwd:1:4:1,c.qkv_t:fwd:1:4:1"
T1all4="$T1w4,t.ffn_down:fwd:1:4:1,t.ffn_up:dgrad:1:4:1,t.qkv:dgrad:1:4:1,c.qkv_t:dgrad:1:4:1"
T1all12="t.ffn_up:fwd:1:12:1,t.ffn_down:dgrad:1:12:1,t.qkv:fwd:1:12:1,c.qkv_t:fwd:1:12:1,t.ffn_down:fwd:1:12:1,t.ffn_up:dgrad:1:12:1,t.qkv:dgrad:1:12:1,c.qkv_t:dgrad:1:12:1,t.out:fwd:1:12:1,t.out:dgrad:1:12:1"
T1all9="t.ffn_up:fwd:1:9:1,t.ffn_down:dgrad:1:9:1,t.qkv:fwd:1:9:1,c.qkv_t:fwd:1:9:1,t.ffn_down:fwd:1:9:1,t.ffn_up:dgrad:1:9:1,t.qkv:dgrad:1:9:1,c.qkv_t:dgrad:1:9:1"
for r in 1 2; do
run "product" ""
run "text phase 1, wide sites on 128x128 (cfg 4)" "$T1w4"
run "text phase 1, every FFN / QKV site on 128x128 (cfg 4)" "$T1all4"
run "text phase 1, every site on 128x64 two stages (cfg 12)" "$T1all12"
run "text phase 1, every FFN / QKV site on 128x128 two stages (cfg 9)" "$T1all9"
run "text phase 1, FFN-up fwd only on cfg 4" "t.ffn_up:fwd:1:4:1"
run "text phase 1, FFN-down dgrad only on cfg 4" "t.ffn_down:dgrad:1:4:1"
run "text phase 1, long-K sites (FFN-down fwd, FFN-up dgrad) on loader waves (cfg 54)" "t.ffn_down:fwd:1:54:1,t.ffn_up:dgrad:1:54:1"
run "text phase 0 (text-only prefix / tail), long-K sites on loader waves (cfg 54)" "t.ffn_down:fwd:0:54:1,t.ffn_up:dgrad:0:54:1,t.qkv:dgrad:0:54:1"
run "text phase 0, every FFN / QKV site on cfg 46 (loader waves 8+4)" "t.ffn_down:fwd:0:46:1,t.ffn_up:dgrad:0:46:1,t.qkv:dgrad:0:46:1,t.ffn_up:fwd:0:46:1,t.ffn_down:dgrad:0:46:1,t.qkv:fwd:0:46:1"
done
