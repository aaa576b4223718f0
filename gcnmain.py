#!/usr/bin/env python
"""Top-level shim keeping the reference's entry point name:
    python gcnmain.py -hid 300 300 300 -bucket 50 -batch 500 -d ./data/cmu -mindf 10 -reg 0.0 -dropout 0.5 -cel 5 -highway
(reference README.md:167) runs the MI355X implementation.  Add `--synthetic cmu|twus` when no dump.pkl exists."""
import sys

from geographconv_amd.gcnmain import *  # noqa: F401,F403
from geographconv_amd.gcnmain import run

if __name__ == '__main__':
    run(sys.argv[1:])
