"""VGPR / scratch / occupancy of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage output).
usage: hipcc <flags of csrc/Makefile> -Rpass-analysis=kernel-resource-usage -c X.hip -o /tmp/x.o 2> /tmp/x_res.txt
       python tools/kernel_regs.py /tmp/x_res.txt <regex on the mangled name>"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split(' ')[0]
    if pat is None or pat.search(name):
        g = lambda k: re.search(k + r': (\d+)', b).group(1)
        print(name, 'VGPR', g('VGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
