cd $GRAFT_REPO_ROOT
bash scripts/var_tvl1_rows.sh "2,512,4,512" "2,512,4,512"
echo "bench: $(timeout -k 10 150 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],2), "ms; update", round(r["avg_launch_us"],1), "forward", round(r["other_kernel"]["avg_launch_us"],1))')"
