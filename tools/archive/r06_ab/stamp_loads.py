import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
t0 = d[:, 2]
f = lambda k: ((d[:, k] - t0) / 100.0)
print("start->image cleared %.2f | ->first symbols of wave 0 arrived %.2f | ->scattered %.2f | ->streamed %.2f | ->decoder input visible %.2f (us, mean over %d workgroups)"
      % (f(7).mean(), f(11).mean(), f(8).mean(), f(9).mean(), f(10).mean(), len(d)))
q = np.argsort(t0); n = len(d) // 4
for k in range(4):
    s = q[k * n:(k + 1) * n]
    print("  start-order quarter %d: cleared %.2f arrived %.2f scattered %.2f" % (k, f(7)[s].mean(), f(11)[s].mean(), f(8)[s].mean()))
