for cfg in 84 82 44 42 43 41; do echo "== ZT_WS_CFG=$cfg"; ZT_WS_CFG=$cfg timeout -k 10 120 python tools/bench_conv.py 2>&1 | grep -E "ws full|residual"; done
