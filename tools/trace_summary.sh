#!/bin/bash
# per-variant summary of tools/trace_s4.py: median cycles of the P1 and P2 half-steps of the first chunk
for lib in quantum-systems_amd/variants/libqs_amd_t*.so; do
  for mode in 2 3; do
    QS_AMD_LIB=$PWD/$lib timeout -k 10 200 python tools/trace_s4.py 55 $mode > /tmp/tr.txt 2>/dev/null
    python3 - "$lib" $mode <<'PY'
import re,sys,statistics
p1=[];p2=[];prev=None;first=None;end=None
rows=[l.split() for l in open('/tmp/tr.txt') if '(+' in l]
names=[' '.join(r[2:]) if r[1].endswith(')') else ' '.join(r[3:]) for r in rows]
ts=[int(r[0]) for r in rows]
for i in range(1,len(rows)):
    d=ts[i]-ts[i-1]; nm=names[i-1]
    if 'P1' in nm: p1.append(d)
    elif 'P2' in nm: p2.append(d)
tot=ts[-1]-ts[0]
print(f"{sys.argv[1].split('_')[-1]:8s} {'(d,c)' if sys.argv[2]=='2' else '(b,a)'}  P1 median {statistics.median(p1):6.0f}  P2 median {statistics.median(p2):6.0f}  wave total {tot} cycles, {len(p1)} steps")
PY
  done
done
