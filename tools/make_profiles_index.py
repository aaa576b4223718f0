#!/usr/bin/env python3
"""profiles/INDEX.json: one machine-readable entry per file under profiles/ -- round, kind, the commit that added it, its size and a
one-line description (the file's own first line / heading, or what its name says).   python tools/make_profiles_index.py"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')

KINDS = [(r'bench.*\.json$', 'bench.py JSON line'), (r'kernel_stats\.md$', 'rocprofv3 --kernel-trace summary (tools/rocprof_summary.py)'),
         (r'pmc.*\.(md|json)$', 'rocprofv3 --pmc summary (tools/pmc_summary.py)'), (r'micro', 'stand-alone micro-benchmark output (tools/micro/)'),
         (r'_ab\.txt$', 'same-box A/B of two builds / settings'), (r'\.md$', 'table / notes'), (r'\.txt$', 'raw tool output'), (r'\.json$', 'JSON record')]


def first_line(path):
    try:
        with open(path, errors='replace') as f:
            for line in f:
                line = line.strip().lstrip('#').strip()
                if line:
                    return line[:240]
    except OSError:
        pass
    return ''


def added_in(rel):
    r = subprocess.run(['git', '-C', ROOT, 'log', '--diff-filter=A', '--format=%h %s', '--', rel], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    return lines[-1][:160] if lines else None


def main():
    out = []
    for name in sorted(os.listdir(P)):
        if name == 'INDEX.json':
            continue
        path = os.path.join(P, name)
        m = re.match(r'r(\d\d)', name)
        kind = next((k for pat, k in KINDS if re.search(pat, name)), 'file')
        desc = first_line(path)
        if name.endswith('.json'):
            try:
                d = json.load(open(path))
                if isinstance(d, dict) and 'ms_per_step' in d:
                    desc = '%.2f ms per step, %s' % (d['ms_per_step'], d.get('config', {}).get('workload', '')[:160])
                elif isinstance(d, dict) and 'state' in d:
                    desc = 'state %s, kernel %s' % (d.get('state'), d.get('kernel'))
            except Exception:
                pass
        out.append({"file": name, "round": int(m.group(1)) if m else None, "kind": kind, "bytes": os.path.getsize(path),
                    "added_in": added_in(os.path.join('profiles', name)), "what": desc})
    json.dump({"how": "tools/make_profiles_index.py (regenerate after adding files)", "files": out}, open(os.path.join(P, 'INDEX.json'), 'w'), indent=1)
    print('%d files indexed' % len(out))


if __name__ == '__main__':
    main()
