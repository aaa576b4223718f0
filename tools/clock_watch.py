#!/usr/bin/env python3
"""Run a command while polling every amdgpu hwmon directory (shader clock, socket power); print, per directory that moved, the
median / min shader clock and the median / max power during the run.  (Measurement tool, not part of the product.)
    python tools/clock_watch.py -- tools/micro/bin/x3_rows loop 0"""
import glob, os, statistics, subprocess, sys, threading, time


def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def main():
    cmd = sys.argv[sys.argv.index('--') + 1:]
    cards = [c for c in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')) if os.path.exists(os.path.join(c, 'freq1_input'))]
    samples = {c: [] for c in cards}
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            for c in cards:
                samples[c].append((rd(os.path.join(c, 'freq1_input')), rd(os.path.join(c, 'power1_input')) or rd(os.path.join(c, 'power1_average'))))
            time.sleep(0.02)

    t = threading.Thread(target=poll)
    t.start()
    rc = subprocess.call(cmd)
    stop.set()
    t.join()
    for c in cards:
        f = [s[0] / 1e6 for s in samples[c] if s[0]]
        p = [s[1] / 1e6 for s in samples[c] if s[1]]
        if not f or not p or max(p) < 400:
            continue
        cap = rd(os.path.join(c, 'power1_cap'))
        print("  [%s] sclk MHz median %.0f min %.0f max %.0f | power W median %.0f max %.0f (cap %s) | %d samples" % (
            c.split('/')[4], statistics.median(f), min(f), max(f), statistics.median(p), max(p), cap / 1e6 if cap else '?', len(f)))
    return rc


if __name__ == '__main__':
    sys.exit(main())
