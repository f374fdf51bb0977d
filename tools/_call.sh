timeout 300 python tools/host_profile2.py 2>&1 | head -24
timeout 300 python tools/shard_regime.py 2>/dev/null | python -c "
import json,sys
o=json.load(sys.stdin)
for k,v in o.items():
    if isinstance(v,dict): print(k, {a:round(b,4) for a,b in v.items() if isinstance(b,float)})
"
