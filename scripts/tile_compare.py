import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print(k, "same bits" if same else f"DIFFERENT: max |d| {np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max():.3e}")
