for cmd in "bench_fir.py --steps 50" "bench_fftfilt.py --steps 50 --no-sweep" "bench_fftfilt.py --steps 50 --no-sweep --taps 4095" "bench_fastddc.py --steps 100" "bench_nfm.py --steps 50"; do
  s=$(date +%s.%N); python $cmd --no-cpu-baseline --verify > /tmp/o.json 2>/tmp/o.err; e=$(date +%s.%N)
  python - <<PY
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print("$cmd", round($e-$s,1), "s", d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("kernel","")[:40], d.get("verify",{}).get("ok"))
PY
done
