"""development: are the arrays of two .npz dumps identical bit for bit?   usage: python tools/npz_equal.py a.npz b.npz [skip-substring ...]"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
skip = sys.argv[3:]
bad = 0
for k in a.files:
    if any(s in k for s in skip): continue
    x, y = a[k], b[k]
    if k.startswith("info"): x, y = x[..., :4], y[..., :4]   # (branch record; iteration statistics may differ between builds)
    if not np.array_equal(x, y):
        bad += 1
        d = np.abs(x.astype(float) - y.astype(float)).max()
        print(f"{k}: differs (max abs {d:.3e})")
print("identical" if not bad else f"{bad} arrays differ", f"({len(a.files)} arrays)")
