import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    out={k:d.get(k) for k in ('value','ms_per_step','steps')}
    v=d.get('roofline',{}).get('valu',{})
    out['valu']={k:(x.get('frac_of_no_fma_bound') if isinstance(x,dict) else x) for k,x in v.items()} if isinstance(v,dict) else v
    out['stage']=d.get('stage_ms_last_step')
    print(json.dumps(out))
