B="python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-traffic --no-train --no-sustain --no-strong"
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1],round(d["value"]),round(d["step_roofline"]["us_per_recurrent_step"],2))'
timeout 200 $B 2>/dev/null | python -c "$P" separate
TNP_CHAIN=1 timeout 200 $B 2>/dev/null | python -c "$P" chain_bk16
TNP_CHAIN=1 TNP_CHAIN_NOWAIT=1 timeout 200 $B 2>/dev/null | python -c "$P" chain_bk16_nowait
TNP_CHAIN=1 TNP_CHAIN_BK=32 timeout 200 $B 2>/dev/null | python -c "$P" chain_bk32_one_wg_per_cu
TNP_CHAIN=1 TNP_CHAIN_BK=32 TNP_CHAIN_NOWAIT=1 timeout 200 $B 2>/dev/null | python -c "$P" chain_bk32_nowait
