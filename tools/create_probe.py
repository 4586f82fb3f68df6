import time, sys, os
sys.path.insert(0, os.getcwd())
t0=time.perf_counter()
from designcsg_amd.evaluator import Evaluator
from designcsg_amd import _lib
_lib.lib()
t1=time.perf_counter()
ev=Evaluator(0)
t2=time.perf_counter()
print(os.environ.get("DCSG_NO_RING_PREALLOC","prealloc"), "create %.4f"%(t2-t1), ev.load_profile()["create"])
ev.close()
