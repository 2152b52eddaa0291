import sys, os, time
sys.path.insert(0, 'tests')
import helpers as H, numpy as np
import svhip as S
print(S.lib().svh_version(), 'devices', S.device_count())
for case in ['urban3_demo', 'urban1_robotics', 'cones_middlebury']:
    z = np.load(os.path.join(H.GOLDEN, case + '.npz'))
    prm = H.ElasParams.from_buffer_copy(z['params'].tobytes())
    l, r = H.golden_pair(str(z['crop']))
    e = S.Elas(prm); e.set_taps(True)
    t = time.time(); rc, D1, D2 = e.process(l, r); print(case, 'rc', rc, 'first call %.1f ms' % ((time.time()-t)*1e3))
    want = H.oracle_elas_run(prm, l, r)
    st = {s: e.stage(s, H.stage_dtype(s)) for s in range(H.STAGE_COUNT)}
    st[H.D1_FINAL] = D1.ravel(); st[H.D2_FINAL] = D2.ravel()
    got = H.StageRun(rc, st)
    print(H.compare_runs(want, got))
    print('agree', H.disparity_agreement(D1.ravel(), z['d1']), H.disparity_agreement(D2.ravel(), z['d2']))
    e2 = S.Elas(prm)
    for i in range(3):
        t = time.time(); e2.process(l, r); dt = (time.time()-t)*1e3
        print('  process %.2f ms' % dt, e2.last_timing())
