#!/usr/bin/env python3
"""python <repo>/dropin/run.py <reference script> [args...]  -- run e.g. jdacs/train.py or jdacs-ms/test.py unchanged with
the four hot-path modules redirected to the MI355X drop-ins (see mvs_dropin.py)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mvs_dropin  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    mvs_dropin.install()
    script = os.path.abspath(sys.argv[1])
    sys.argv = sys.argv[1:]
    sys.path[0] = os.path.dirname(script)     # what `python script.py` would have put there
    runpy.run_path(script, run_name="__main__")
