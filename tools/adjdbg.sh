for d in 0 1 2 3; do echo "== AGCN_AF_DBG=$d"; AGCN_AF_DBG=$d REPS=10 timeout -k 10 200 python tools/bench_adj.py l2 l6 l9; done
