import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64)[:11].astype(float)
n = max(a[10], 1)
print("per chunk (us): at B0 %.3f | at the level barriers %s | rest of the chunk %.3f | chunks %d" % (a[0] * 0.01 / n, " ".join("%.3f" % (v * 0.01 / n) for v in a[1:9]), a[9] * 0.01 / n, a[10]))
