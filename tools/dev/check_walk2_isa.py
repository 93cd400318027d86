#!/usr/bin/env python3
"""CLI over 3dgp_amd/isa_check.py (the check itself runs inside `3dgp_amd/build.py` on every rebuild of field.hip):
    python tools/dev/check_walk2_isa.py [extra hipcc flags...]     compiles 3dgp_amd/csrc/field.hip to assembly with the build's flags and
                                                                   checks every triplane_walk2_kernel instantiation (exit 1 on a violation)"""
import importlib
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module('3dgp_amd.build')
isa = importlib.import_module('3dgp_amd.isa_check')

with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'f.s')
    subprocess.check_call([pkg._hipcc()] + pkg.FLAGS + sys.argv[1:] + ['-S', '--cuda-device-only', '-o', out, os.path.join(pkg.CSRC, 'field.hip')],
                          stderr=subprocess.DEVNULL)
    res = isa.check_walk2_asm(open(out).read())
rc = 0
for name, (summary, bad) in res.items():
    print(name, summary)
    for why, ins in bad[:20]:
        print('  VIOLATION:', why, '|', ins)
        rc = 1
sys.exit(rc)
