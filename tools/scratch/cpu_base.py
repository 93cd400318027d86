import sys, os, importlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
t = importlib.import_module('3dgp_amd')
print(json.dumps(bench.cpu_baseline(t, t.config.config_c3())))
