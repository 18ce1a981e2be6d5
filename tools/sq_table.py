"""Developer: one line per kernel from a tools/pmc_summary.py file (gpurun_out/<tag>_sq.txt)."""
import sys
for path in sys.argv[1:]:
    cur = None
    d = {}
    for l in open(path):
        if not l.startswith(' '):
            cur = l.strip(); d[cur] = {}
        else:
            p = l.split(); d[cur][p[0]] = float(p[1])
    print("==", path)
    tot = 0.0
    for k, v in d.items():
        if 'SQ_WAVES' not in v or v['SQ_WAVES'] < 200:
            continue
        w = v['SQ_WAVES']; tot += v['SQ_INSTS_VALU']
        print(f"{k:24s} waves {w:6.0f} valu/w {v['SQ_INSTS_VALU']/w:6.0f} salu/w {v['SQ_INSTS_SALU']/w:5.0f} vmrd/w {v.get('SQ_INSTS_VMEM_RD',0)/w:4.0f} vmwr/w {v.get('SQ_INSTS_VMEM_WR',0)/w:4.0f} lds/w {v.get('SQ_INSTS_LDS',0)/w:4.0f} "
              f"cyc/w {4*v['SQ_WAVE_CYCLES']/w:7.0f} wait% {100*v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:3.0f} stall% {100*v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:3.0f} act% {100*v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']:3.0f} totVALU {v['SQ_INSTS_VALU']/1e6:5.1f}M")
    print(f"total VALU {tot/1e6:.1f}M")
