"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into the text files committed under profiles/.

  python tools/rocpd_summary.py stats <results.db>          # per-kernel calls / total / avg / min / max (what --stats reports)
  python tools/rocpd_summary.py pmc <results.db> [...]      # per-kernel mean of every collected counter
  python tools/rocpd_summary.py traffic <out.json> <results.db> [...]   # FETCH_SIZE / WRITE_SIZE per launch of the network kernels (bench.py reads it)
"""
import sqlite3
import sys


def short(name):
  name = name.replace('void ', '')
  return name if len(name) < 70 else name[:67] + '...'


def stats(path):
  cur = sqlite3.connect(path).cursor()
  rows = list(cur.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), '
                          'max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) '
                          'from kernels group by name order by sum(duration) desc'))
  tot = sum(r[2] for r in rows)
  print(f'# rocprofv3 --kernel-trace --stats   ({path})   durations in microseconds')
  print(f'{"kernel":70s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>7s} {"scratch":>7s} {"grid":>9s} {"wg":>5s}')
  for n, c, s, a, mn, mx, vg, ag, sg, lds, scr, gx, wx in rows:
    print(f'{short(n):70s} {c:6d} {s / 1e3:12.1f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / tot:6.2f} {vg:5d} {ag:5d} {sg:5d} {lds:7d} {scr:7d} {gx:9d} {wx:5d}')


def pmc(paths):
  print('# rocprofv3 --pmc ... (separate passes)   per-kernel mean counter value per dispatch (summed over XCDs/SEs as rocprofv3 reports it)')
  for path in paths:
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                            'group by kernel_name, counter_name order by kernel_name, counter_name'))
    print(f'## {path}')
    print(f'{"kernel":70s} {"counter":28s} {"dispatches":>10s} {"mean_value":>16s} {"mean_dur_us":>12s}')
    for n, cn, c, v, d in rows:
      if n.startswith('k_') or n.startswith('void k_'):
        print(f'{short(n):70s} {cn:28s} {c:10d} {v:16.1f} {d / 1e3:12.2f}')


def traffic(out, paths):
  import json
  res = {}
  for path in paths:
    cur = sqlite3.connect(path).cursor()
    for n, cn, c, v in cur.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection '
                                   "where counter_name in ('FETCH_SIZE', 'WRITE_SIZE') group by kernel_name, counter_name"):
      key = n.replace('void ', '').split('<')[0].split('(')[0]
      res.setdefault(key, {'source': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-x6'})
      res[key][cn + '_KB'] = v
      res[key][cn + '_dispatches'] = c
  with open(out, 'w') as f:
    json.dump(res, f, indent=1, sort_keys=True)
  print(json.dumps(res.get('k_static_views'), indent=1))


if __name__ == '__main__':
  if sys.argv[1] == 'stats':
    stats(sys.argv[2])
  elif sys.argv[1] == 'traffic':
    traffic(sys.argv[2], sys.argv[3:])
  else:
    pmc(sys.argv[2:])
