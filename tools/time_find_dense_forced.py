"""word soup forced through the selection kernels (find3 = left3 = 2): what the dense-text gate avoids — python tools/time_find_dense_forced.py"""
import sys, runpy
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import daachorse_amd as da
da.set_option("find3", 2); da.set_option("left3", 2)
for what in ("find", "leftmost"):
    sys.argv = ["time_find.py", "1024", "dense", what]
    runpy.run_path(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "time_find.py"), run_name="__main__")
