"""Development only: run a script of this repo against a VARIANT build of the same C ABI (tools/dev/build_variant.sh).
   python tools/dev/with_lib.py tools/dev/variants/x.so bench.py --steps 10 ...
The product has no such switch: 3dgp_amd/_lib.py loads 3dgp_amd/csrc/libtdgp_hip.so and nothing else."""
import importlib
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
so, script = sys.argv[1], sys.argv[2]
if so != 'default':
    importlib.import_module('3dgp_amd')._lib.LIB_PATH = os.path.abspath(so)
sys.argv = sys.argv[2:]
runpy.run_path(os.path.join(REPO, script) if not os.path.isabs(script) else script, run_name='__main__')
