import numpy as np, sys
M, KD, N = 32768, 256, 128
def rd(p):
    b = open(p, "rb").read()
    C = np.frombuffer(b[:M*N*4], np.float32).reshape(M, N); o = M*N*4
    dw = np.frombuffer(b[o:o+KD*N*8], np.float64).reshape(KD, N); o += KD*N*8
    s = np.frombuffer(b[o:o+2*N*8], np.float64).reshape(2, N)
    return C, dw, s
a, b = rd(sys.argv[1]), rd(sys.argv[2])
for n, x, y in zip(("dY", "dW", "stats"), a, b):
    d = np.abs(x.astype(np.float64) - y)
    print(n, "max|ref| %.3e  max diff %.3e  mean diff %.3e" % (np.abs(x).max(), d.max(), d.mean()))
C0, C1 = a[0], b[0]
bad = np.abs(C0 - C1) > 1e-3 * np.abs(C0).max()
print("bad elements", bad.sum(), "of", bad.size)
if bad.any():
    r, c = np.nonzero(bad)
    print("rows mod 64 histogram", np.bincount(r % 64, minlength=64))
    print("cols histogram", np.bincount(c, minlength=128))
    print("first bad", r[:8], c[:8], C0[r[:8], c[:8]], C1[r[:8], c[:8]])
