for cfg in 0 1 2 4 8; do set -- $cfg; echo "children of the first $1 nodes of a launch: $(MPLX_PLAN_SPEC=$1 python profiles/micro/plan_c1_once.py 2>&1 | tail -1)"; done
