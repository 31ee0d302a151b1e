import sys, runpy, os
sys.path.insert(0, os.getcwd())
import hite_amd._lib as L
L.SO_PATH = os.path.join(os.path.dirname(L.__file__), sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
