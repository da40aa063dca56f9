import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import faulthandler; faulthandler.enable()
from test_streams_and_graph import run
multi, graph = int(sys.argv[1]), int(sys.argv[2])
p, s = run(bool(multi), bool(graph))
print('ok', multi, graph, float(abs(p).sum()))
