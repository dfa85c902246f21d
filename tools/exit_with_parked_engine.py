"""A process that ends while run_dream still holds a parked engine (restart=True would continue on it): the engine is released at
interpreter exit, before the HIP runtime goes away -- rc 0."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pydream_amd.core import run_dream
from pydream_amd.parameters import FlatParam
from pydream_amd.likelihoods import MVNormalLogLike
from tests import helpers as H
os.chdir(tempfile.mkdtemp())
d, N = 8, 8
Z0 = H.seed_history(80, d, 1); np.save("s.npy", Z0)
s, _ = run_dream([FlatParam(np.zeros(d))], MVNormalLogLike(H.mvn_precision(d)), nchains=N, niterations=40, start=[Z0[i] for i in range(N)], history_file="s.npy",
                 model_name="m", save_history=True, verbose=False, multitry=5)
print("leaving with a parked engine")
