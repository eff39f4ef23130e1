"""Per-kernel table of the training iteration from the PMC summary tools/prof_train.sh writes (train_full_pmc.txt): mean HBM traffic per
launch (FETCH_SIZE x 2 -- the gfx950 half-count correction of MI355X_MICROARCH.md -- + WRITE_SIZE), mean launch duration, achieved TB/s
against the 8 TB/s HBM peak, and the matrix pipe's busy fraction SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
usage: python tools/train_rooflines.py gpurun_out/prof_train/train_full_pmc.txt > profiles/rNN_train_rooflines.txt"""
import re
import sys

src = sys.argv[1]
rows = {}
for line in open(src):
  m = re.match(r'^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
  if not m:
    continue
  k, c, n, v, dur = m.group(1), m.group(2), int(m.group(3)), float(m.group(4)), float(m.group(5))
  r = rows.setdefault(k, {})
  r[c], r['n'] = v, n
  if c == 'FETCH_SIZE':
    r['dur'] = dur
print('# Training iteration (tools/trainbench.py 3072 full, 5 timed iterations + warm-up): per-kernel HBM traffic from rocprofv3 PMC passes')
print('# (FETCH_SIZE x 2 [gfx950 half-count correction] + WRITE_SIZE, counters are KiB, mean per launch),')
print('# mean launch duration, achieved TB/s against the 8 TB/s HBM peak, and the matrix pipe busy fraction')
print('# (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)).  Source: %s' % src.split('/')[-1])
print('%-60s %9s %9s %10s %6s %9s %9s %8s' % ('kernel', 'launches', 'avg_us', 'MB/launch', 'TB/s', 'of 8 TB/s', 'MFMA busy', 'ms total'))
out = []
for k, r in rows.items():
  if 'FETCH_SIZE' not in r or 'WRITE_SIZE' not in r:
    continue
  by = (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0  # the counters are in KiB
  tbs = by / (r['dur'] * 1e-6) / 1e12 if r['dur'] > 0 else 0.0
  busy = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (r['GRBM_GUI_ACTIVE'] / 8 * 1024) if r.get('GRBM_GUI_ACTIVE') else 0.0
  out.append((r['n'] * r['dur'], k, r['n'], r['dur'], by / 1e6, tbs, tbs / 8, busy))
for tot, k, n, dur, mb, tbs, frac, busy in sorted(out, reverse=True):
  print('%-60s %9d %9.1f %10.1f %6.2f %9.2f %9.2f %8.1f' % (k[:60], n, dur, mb, tbs, frac, busy, tot / 1e3))
