import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import test_dist_graph_gpu as T
for mode, port in (("single", 29701), ("single", 29702), ("eager", 29703), ("eager", 29704), ("graph", 29705)):
    r = T.run(mode, port)
    print(mode, ["%.9f" % v for v in r["losses"]], "%.12e" % r["w"])
