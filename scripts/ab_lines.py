import json,sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); r=d['roofline']
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'conv', r['all_sparse_conv_ms_per_step_warmup'], 'pipe', d.get('pipelined',{}).get('value'), 'avg128', r['avg_launch_us'], [ (k['kernel'].split('(')[0][-28:], k['ms_per_step']) for k in r['kernels_warmup'][:3]])
