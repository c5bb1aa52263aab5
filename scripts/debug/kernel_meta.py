import re,sys
s=open(sys.argv[1]).read()
blocks = re.split(r'\n  - \.agpr_count:', s)
for b in blocks[1:]:
    ag = re.match(r'\s*(\d+)', b).group(1)
    g=lambda k: re.search(r'\.'+k+r':\s+(\S+)', b).group(1)
    print(g('name'), 'agpr',ag,'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'vspill',g('vgpr_spill_count'),'sspill',g('sgpr_spill_count'),'scratch',g('private_segment_fixed_size'))
