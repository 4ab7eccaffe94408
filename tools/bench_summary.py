import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    print(f.split('/')[-1], d['config']['workload'][:30], 'value',d['value'],'ok',d['roundtrip_ok'],'kms',d['kernel_ms'],'frac',d['hbm_frac'])
    for c in d.get('configs',[]):
        print('   ',c['workload'],'value',c['value'],'ok',c['roundtrip_ok'],'kms',c['kernel_ms'],'frac',c['hbm_frac'])
