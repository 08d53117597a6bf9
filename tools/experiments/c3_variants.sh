for v in 0 5376 24 5400 25 5401; do
python bench.py --config directional --scenes 256 --steps 40 --warmup 5 --no-cpu-baseline --no-traffic --no-train --no-sustain --no-strong --no-roofline --variant $v 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('variant',$v,round(d['value']),round(d['ms_per_step'],3))"
done
