"""Tabulate hipcc -Rpass-analysis=kernel-resource-usage output:  python tools/resusage.py <stderr file>"""
import re, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
  name = b.split()[0]
  g = lambda k: re.search(re.escape(k) + r': (\d+)', b).group(1)
  print('%-50s sgpr %4s vgpr %4s agpr %4s scratch %4s occ %s' % (name[:48], g('TotalSGPRs'), g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]')))
