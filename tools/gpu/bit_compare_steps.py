import os, sys, subprocess, tempfile, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/tools')
import bit_compare as B
a, b = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    ra, rb = B.run(a, 'unitree_go2_trot', 64, 4, td + '/a.npz'), B.run(b, 'unitree_go2_trot', 64, 4, td + '/b.npz')
    for k in ('qss', 'qdss', 'xss', 'rewss'):
        x, y = ra[k], rb[k]
        print(k, x.shape)
        d = (x.view(np.uint32) != y.view(np.uint32))
        if d.ndim == 3:
            for t in range(min(3, d.shape[1])):
                print('  step', t, 'rollouts differing', int(d[:, t].any(axis=-1).sum()), 'columns', np.nonzero(d[:, t].any(axis=0))[0][:40], 'max', float(np.abs(x[:, t] - y[:, t]).max()))
        else:
            print('  ', d.sum(axis=0)[:6])
