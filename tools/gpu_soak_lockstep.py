"""Soak of the pipelined lockstep loop: K = 16 objects through svh_vo_prefetch_batch / svh_vo_process_next_batch for N seconds,
object 0 compared with a single VisualOdometryStereo object on the same frames after every call (return value, motion,
inliers), host RSS and free device memory at call 200 and at the end.   python tools/gpu_soak_lockstep.py [seconds]"""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def gpu_free():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value
im = [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
K = 16
prm = H.vo_defaults()
vos = [H.ProductVo(prm, private_rand=0) for _ in range(K)]
ref = H.ProductVo(prm, private_rand=0)
seq = [[np.roll(a, 3 * k, axis=1) for a in im] for k in range(K)]
even = ([s[0] for s in seq], [s[1] for s in seq]); odd = ([s[2] for s in seq], [s[3] for s in seq])
shape = im[0].shape
H.product_vo_prefetch_batch(vos, *even)
t0 = time.time(); i = 0; rss0 = free0 = None
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 40:
    nxt = odd if i % 2 == 0 else even
    n, ok = H.product_vo_process_next_batch(vos, nxt[0], nxt[1], shape)
    a, b = (even if i % 2 == 0 else odd)
    r = ref.process(a[0], b[0])            # object 0 alone, same frames: must stay identical
    assert r == ok[0] and np.array_equal(ref.motion(), vos[0].motion()) and np.array_equal(ref.inliers(), vos[0].inliers()), i
    i += 1
    if i == 200:
        rss0, free0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, gpu_free()
rss1, free1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, gpu_free()
print("soak: %d pipelined lockstep calls x %d objects, object 0 identical to a single object throughout; host max RSS %d -> %d KB, device free %d -> %d MB"
      % (i, K, rss0, rss1, free0 >> 20, free1 >> 20))
