"""Loop instruction histogram of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

  python tools/isa_loops.py file.s <mangled-name-regex> [top]

Prints every backward branch (loop) with its length, the opcode histogram of the longest one and the kernel's
register / scratch figures from the .amdhsa_kernel block.
"""
import collections
import re
import sys


def main():
  path, pat = sys.argv[1], sys.argv[2]
  top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
  txt = open(path).read()
  lines = txt.split('\n')
  starts = [i for i, l in enumerate(lines) if re.match(r'^(%s)\w*:' % pat, l)]
  for start in starts:
    name = lines[start].split(':')[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
      m = re.match(r'^(\.LBB\d+_\d+):', l)
      if m:
        labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
      m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
      if m:
        t = m.group(1) or m.group(2)
        if labels.get(t, 1 << 30) < i:
          loops.append((labels[t], i))
    k = txt.find('.amdhsa_kernel ' + name)
    seg = txt[k:k + 6000]
    regs = {key: re.search(r'\.amdhsa_%s (\d+)' % key, seg) for key in ('next_free_vgpr', 'next_free_sgpr', 'accum_offset')}
    scratch = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', seg)
    print(name[:90])
    print('  ', {k2: int(v.group(1)) for k2, v in regs.items() if v}, 'scratch', scratch.group(1) if scratch else '?',
          'instructions', sum(1 for l in body if l.strip() and not l.strip().startswith(('.', ';')) and not l.endswith(':')))
    for a, b in loops:
      print('   loop lines %d..%d (%d)' % (a, b, b - a))
    if loops:
      a, b = max(loops, key=lambda ab: ab[1] - ab[0])
      cnt = collections.Counter()
      for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
          continue
        cnt[l.split()[0]] += 1
      print('   longest loop:', sum(cnt.values()), 'instructions:', ', '.join('%s %d' % kv for kv in cnt.most_common(top)))


if __name__ == '__main__':
  main()
