#!/bin/bash
# A/B on one box: alternate two environments (given as "VAR=val" strings, "-" for none) over bench.py; prints step ms
# and the per-kernel event times.  Usage: tools/ab_bench.sh "SRFRD_NO_LSPEC=1" "-" [rounds]
A="$1"; B="$2"; R="${3:-2}"
for r in $(seq 1 $R); do
  for V in "$A" "$B"; do
    if [ "$V" = "-" ]; then E=""; else E="$V"; fi
    out=$(env $E timeout -k 10 200 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null)
    echo "[$V] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print("ms_per_step=%.4f"%d["ms_per_step"], " ".join("%s=%.1f"%(n.replace("srfrd_",""),v*1000) for n,v in k.items()))')"
  done
done
