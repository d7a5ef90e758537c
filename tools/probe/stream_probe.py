import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mesheditor_amd import api
from tools import lab
ctx = api.Context(0)
for nbytes in (1 << 30, 4 << 30):
    c, r = lab.bench_stream(ctx, nbytes, 10)
    print("streaming over %d GiB: copy %.0f GB/s (read + written bytes), read %.0f GB/s" % (nbytes >> 30, c, r))
