import re, subprocess, sys
s=open(sys.argv[1]).read()
pats=sys.argv[2:]
blocks=s.split('- .agpr_count')
for b in blocks[1:]:
    name=re.search(r'\.name:\s+(\S+)',b).group(1)
    g=lambda k: (re.search(r'\.%s:\s+(\d+)'%k,b) or [None,'?'])[1]
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    if not pats or any(k in dn for k in pats):
        print('%-100s vgpr %s sgpr %s spill %s lds %s scratch %s'%(dn[:100],g('vgpr_count'),g('sgpr_count'),g('vgpr_spill_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
