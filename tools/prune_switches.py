"""One-off source maintenance (round 5, review item 8): resolve the preprocessor conditionals of csrc that depend ONLY on A/B switches whose experiments are closed,
leaving every other conditional untouched.   python tools/prune_switches.py FILE... (edits in place; the switch table is below)"""
import itertools
import re
import sys

KNOWN = {'B6_SPREAD': 0, 'B6_PINNED': 0, 'B6_PRIO_STATIC': 0, 'B6_ASM_DMA': 0, 'B6_SCHED': 0, 'B6_VALU_PER_PAIR': 1, 'DYN_ENGINE_B6': 1}


def truth(expr):
  """-> True / False if the expression's value does not depend on anything but KNOWN, else None."""
  e = re.sub(r'//.*', '', expr)
  e = re.sub(r'/\*.*?\*/', '', e)
  unknown = []

  def sub_defined(m):
    n = m.group(1) or m.group(2)
    if n in KNOWN:
      return '1'
    unknown.append('defined:' + n)
    return f'U[{len(unknown) - 1}]'

  e = re.sub(r'defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)', sub_defined, e)

  def sub_ident(m):
    n = m.group(0)
    if n == 'U':
      return n
    if n in KNOWN:
      return str(KNOWN[n])
    unknown.append(n)
    return f'U[{len(unknown) - 1}]'

  e = re.sub(r'\b[A-Za-z_]\w*\b', sub_ident, e)
  e = e.replace('&&', ' and ').replace('||', ' or ')
  e = re.sub(r'!(?!=)', ' not ', e)
  if not any(k in expr for k in KNOWN):
    return None
  vals = set()
  for combo in itertools.product((0, 1, 7), repeat=len(unknown)):
    try:
      vals.add(bool(eval(e, {'U': list(combo)})))
    except Exception:
      return None
  return vals.pop() if len(vals) == 1 else None


def prune(lines):
  out = []
  # stack entries: dict(mode='resolved'|'keep', emitting=bool, taken=bool, parent_emitting=bool)
  stack = []
  emitting = lambda: all(s['emitting'] for s in stack)
  i = 0
  while i < len(lines):
    l = lines[i]
    s = l.strip()
    m = re.match(r'#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)', s)
    if not m:
      if emitting():
        out.append(l)
      i += 1
      continue
    kind, rest = m.group(1), m.group(2).strip()
    if kind in ('if', 'ifdef', 'ifndef'):
      if kind == 'ifndef' and rest.split()[0] in KNOWN and i + 2 < len(lines) and lines[i + 1].strip().startswith('#define ' + rest.split()[0]) and lines[i + 2].strip().startswith('#endif'):
        i += 3  # the default definition of a pruned switch
        continue
      t = truth(rest) if kind == 'if' else None
      if t is None:
        stack.append(dict(mode='keep', emitting=True))
        if emitting():
          out.append(l)
      else:
        stack.append(dict(mode='resolved', emitting=t, taken=t))
    elif kind == 'elif':
      top = stack[-1]
      if top['mode'] == 'keep':
        if emitting():
          out.append(l)
      else:
        if top['taken']:
          top['emitting'] = False
        else:
          t = truth(rest)
          if t is None:
            raise SystemExit(f'line {i + 1}: #elif with an open condition behind a resolved #if: resolve by hand')
          top['emitting'] = t
          top['taken'] = t
    elif kind == 'else':
      top = stack[-1]
      if top['mode'] == 'keep':
        if emitting():
          out.append(l)
      else:
        top['emitting'] = not top['taken']
        top['taken'] = True
    else:  # endif
      top = stack.pop()
      if top['mode'] == 'keep' and emitting():
        out.append(l)
    i += 1
  assert not stack
  return out


for path in sys.argv[1:]:
  src = open(path).read().split('\n')
  new = prune(src)
  open(path, 'w').write('\n'.join(new))
  print(path, len(src), '->', len(new), 'lines')
