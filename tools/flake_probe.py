import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydream_amd.core import run_dream
from pydream_amd.likelihoods import MVNormalLogLike
from pydream_amd.parameters import FlatParam
from tests import helpers as H
d, N, n = 10, 6, 40
P = H.mvn_precision(d)
dev = MVNormalLogLike(P, factorize=False)
host = lambda x: -.5 * np.sum(x * np.dot(P, x))
hist = "/tmp/_probe_seed.npy"
np.save(hist, H.seed_history(100, d, 8))
kw = dict(nchains=N, niterations=n, verbose=False, save_history=False, history_file=hist, multitry=5, seed=6,
          start=[H.seed_history(100, d, 8)[i] for i in range(N)])
ref = None
for mode in sys.argv[1:]:
    os.environ["DZ_COOP"] = "0" if "nocoop" in mode else "1"
    like = host if mode.startswith("host") else dev
    outs = []
    for rep in range(12):
        s, l = run_dream([FlatParam(np.zeros(d))], like, **kw)
        outs.append(np.array(s))
    if ref is None:
        ref = outs[0]
    diff = [int(np.argwhere(np.any(np.abs(o - ref) > 1e-9 * np.abs(ref) + 1e-12, axis=(0, 2)))[0][0]) if np.any(np.abs(o - ref) > 1e-9 * np.abs(ref) + 1e-12) else -1 for o in outs]
    print(mode, "first differing iteration per repetition (-1 = equal to the reference run):", diff)
