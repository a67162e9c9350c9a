import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('pipelined',{}).get('value'))
