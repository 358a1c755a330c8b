"""bench.py against an experimental build of the library:  FRCNN_EXP_LIB=tools/bin/x.so python tools/bench_exp.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib as _l
if os.environ.get("FRCNN_EXP_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
import bench
bench.main()
