"""A user-written device likelihood: the d-dimensional "banana" (twisted Gaussian of Haario et al. 1999) as a HIP kernel.

The reference takes any Python callable as the likelihood (pydream/model.py:17-32) and its examples write theirs in numpy
(pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52).  Here the same density is written twice -- `banana_host`, the Python
callable a PyDREAM user would write, and `SOURCE`, its HIP twin -- and handed to run_dream as a DeviceKernelLogLike: the kernel is launched
once per batch of proposals where the built-in densities' kernels run (include/dreamzs.h dz_set_likelihood_module), so thousands of
chains x tries are evaluated in lockstep on the device instead of through the host callback.

    python -m pydream_amd.examples.banana.banana_device
"""
import numpy as np

B = 0.1          # the twist

SOURCE = r"""
// log L(x) = -1/2 ( x0^2 / 100 + (x1 + b x0^2 - 100 b)^2 + sum_{j >= 2} x_j^2 ),  b = data[0]; one thread per point
extern "C" __global__ void banana_logp(const double* X, long long n, int d, int ld, double* like, const void* data)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* x = X + i * ld;
    const double b = ((const double*)data)[0];
    const double x0 = x[0], x1 = x[1];
    const double t = (x1 + b * (x0 * x0)) - 100.0 * b;
    double acc = (x0 * x0) / 100.0;
    acc = acc + t * t;
    for (int j = 2; j < d; ++j) acc = acc + x[j] * x[j];
    like[i] = -0.5 * acc;
}
"""


def banana_host(x, b=B):
    """the same expression, operation for operation (sequential sum over the dimensions): bit-identical to the kernel"""
    x = np.asarray(x, dtype=float)
    t = (x[1] + b * (x[0] * x[0])) - 100.0 * b
    acc = (x[0] * x[0]) / 100.0
    acc = acc + t * t
    for j in range(2, len(x)):
        acc = acc + x[j] * x[j]
    return -0.5 * acc


def banana_host_batch(X, b=B):
    """[n, d] -> (prior[n] = 0, like[n]); vectorised over the points, sequential over the dimensions"""
    X = np.asarray(X, dtype=float)
    t = (X[:, 1] + b * (X[:, 0] * X[:, 0])) - 100.0 * b
    acc = (X[:, 0] * X[:, 0]) / 100.0
    acc = acc + t * t
    for j in range(2, X.shape[1]):
        acc = acc + X[:, j] * X[:, j]
    return np.zeros(len(X)), -0.5 * acc


def make_likelihood(ndim):
    from pydream_amd.likelihoods import DeviceKernelLogLike
    return DeviceKernelLogLike("banana_logp", ndim, source=SOURCE, data=np.array([B]), always_finite=True, host=banana_host)


if __name__ == "__main__":
    import time
    from pydream_amd.core import run_dream
    from pydream_amd.convergence import Gelman_Rubin
    from pydream_amd.parameters import SampledParam
    from scipy.stats import norm
    d, nchains, niter = 10, 1024, 2000
    like = make_likelihood(d)
    params = [SampledParam(norm, loc=np.zeros(d), scale=np.full(d, 50.0))]          # a wide normal prior, evaluated on the device as well
    t0 = time.time()
    sampled, log_ps = run_dream(params, like, nchains=nchains, niterations=niter, multitry=5, nseedchains=2 * nchains, save_history=False, verbose=False)
    dt = time.time() - t0
    print("%d chains x %d iterations x %d-D banana through a user kernel: %.2f s (%.1f M proposals/s incl. set-up and download); max R-hat %.3f"
          % (nchains, niter, d, dt, nchains * 5 * niter / dt / 1e6, float(np.max(Gelman_Rubin(sampled)))))
    x = np.concatenate([s[niter // 2:] for s in sampled[:64]])
    print("second-half sample mean of x0, x1: %.3f %.3f (truth 0, 0); var x0 %.1f (truth 100)" % (x[:, 0].mean(), x[:, 1].mean(), x[:, 0].var()))
