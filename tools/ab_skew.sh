for rep in 1 2 3; do for skew in 0 4352 69632 1052672; do echo -n "skew $skew: "; HK_BENCH_SKEW=$skew python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity $1 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'])"; done; done
