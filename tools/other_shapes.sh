#!/bin/bash
# GPU box: bench lines of the secondary shapes (config 1: 32^4 m=8 S=1; wide block: 64^3x32 m=32 S=8)
python bench.py --no-cpu-baseline --steps 50 --warmup 5 --local-dims 32 32 32 32 --m 8 --shifts 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=d['steps']; print('32^4 m=8 S=1:', round(d['ms_per_step'],3), 'ms', round(d['hbm_frac_of_bytes_actually_moved'],3), {k: round(v/n,3) for k,v in d['kernel_ms'].items()})"
python bench.py --no-cpu-baseline --steps 6 --warmup 2 --local-dims 64 64 64 32 --m 32 --shifts 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=d['steps']; print('64^3x32 m=32 S=8:', round(d['ms_per_step'],3), 'ms', round(d['hbm_frac_of_bytes_actually_moved'],3), {k: round(v/n,3) for k,v in d['kernel_ms'].items()})"
