import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
print("before build: avail", torch.cuda.is_available())
g.build()
from skrec import _hip
print("after build: avail", torch.cuda.is_available(), "count", _hip.lib().skr_device_count(), _hip.lib().skr_last_error())
g.smoke()
