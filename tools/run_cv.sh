for e in "X=0" "X=1" "X=2"; do
echo "=== $e: $(env $e python bench.py --no_cpu_baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step'],2), round(r['avg_launch_us'],1), round(r['bwd_avg_launch_us'],1), r.get('min_launch_us'))")"
done
rocm-smi --showclocks 2>/dev/null | head -20
