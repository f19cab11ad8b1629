import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import faulthandler; faulthandler.enable()
import hpgmg_amd as H
from hpgmg_testlib import Backend
from test_gpu_operators import VARIANTS
hip = Backend.hip()
hip.configure(**VARIANTS["7pt-cheby-helm"])
lv = hip.level(1, 4)
L = hip.lib
def say(s): print(s, flush=True)
say("level ok")
L.add_vectors(lv.ptr, 0, 0.75, 1, -1.25, 2); say("add")
L.mul_vectors(lv.ptr, 3, 2.0, 1, 2); say("mul")
L.hpgmg_operators_flush(); say("flush")
L.invert_vector(lv.ptr, 4, 3.0, 5); say("invert")
L.scale_vector(lv.ptr, 1, -0.5, 2); say("scale")
say(L.norm(lv.ptr, 3))
