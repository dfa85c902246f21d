"""A user-written device likelihood: the d-dimensional "banana" (twisted Gaussian of Haario et al. 1999) as a HIP kernel.

The reference takes any Python callable as the likelihood (pydream/model.py:17-32) and its examples write theirs in numpy
(pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52).  Here the same density is written twice -- `banana_host`, the Python
callable a PyDREAM user would write, and `SOURCE`, its HIP twin -- and handed to run_dream as a DeviceKernelLogLike: the kernel is launched
once per batch of proposals where the built-in densities' kernels run (include/dreamzs.h dz_set_likelihood_module), so thousands of
chains x tries are evaluated in lockstep on the device instead of through the host callback.  `FUNCTION_SOURCE` is the same density as a
wave-level device function (DeviceFunctionLogLike): compiled into the persistent generation kernel itself.

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


# The same density as a wave-level DEVICE FUNCTION (pydream_amd.likelihoods.DeviceFunctionLogLike): all 64 lanes of a wave evaluate one point --
# lane l adds the squares of its dimensions l, l + 64, ... (the first two dimensions carry the twist), dz_wave_sum adds the lanes' sums by the xor
# butterfly -- and the function is compiled INTO the persistent generation kernel, so the generations run at that kernel's rate instead of the
# multi-kernel path's (a different order of additions than `SOURCE`: it has its own Python twin).
FUNCTION_SOURCE = r"""
__device__ double banana_wave(const double* x, int d, const void* data, int lane)
{
    const double b = ((const double*)data)[0];
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) {
        double term = x[j] * x[j];
        if (j == 0) term = term / 100.0;
        if (j == 1) { const double t = (x[1] + b * (x[0] * x[0])) - 100.0 * b; term = t * t; }
        acc = acc + term;
    }
    return -0.5 * dz_wave_sum(acc);
}
"""


def banana_host_wave(x, b=B):
    """FUNCTION_SOURCE operation for operation: 64 partial sums over strided dimensions, then the xor butterfly 32, 16, ..., 1"""
    x = np.asarray(x, dtype=float)
    part = np.zeros(64)
    for j in range(len(x)):
        term = x[j] * x[j]
        if j == 0:
            term = term / 100.0
        if j == 1:
            t = (x[1] + b * (x[0] * x[0])) - 100.0 * b
            term = t * t
        part[j % 64] = part[j % 64] + term
    for o in (32, 16, 8, 4, 2, 1):
        part = part + part[np.arange(64) ^ o]
    return -0.5 * part[0]


def make_function_likelihood(ndim):
    from pydream_amd.likelihoods import DeviceFunctionLogLike
    return DeviceFunctionLogLike(FUNCTION_SOURCE, "banana_wave", ndim, data=np.array([B]), always_finite=True, host=banana_host_wave)


if __name__ == "__main__":
    import time
    from pydream_amd.core import run_dream
    from pydream_amd.convergence import Gelman_Rubin
    from pydream_amd.parameters import SampledParam
    from scipy.stats import norm
    d, nchains, niter = 10, 1024, 2000
    params = [SampledParam(norm, loc=np.zeros(d), scale=np.full(d, 50.0))]          # a wide normal prior, evaluated on the device as well
    for what, like in (("a user batch kernel (multi-kernel path)", make_likelihood(d)), ("a user device function inside the persistent kernel", make_function_likelihood(d))):
        like.code_object()                                                          # (compiled on first use: keep hipcc out of the timing)
        t0 = time.time()
        sampled, log_ps = run_dream(params, like, nchains=nchains, niterations=niter, multitry=5, nseedchains=2 * nchains, save_history=False, verbose=False)
        dt = time.time() - t0
        print("%d chains x %d iterations x %d-D banana through %s: %.2f s (%.1f M proposals/s incl. set-up and download); max R-hat %.3f"
              % (nchains, niter, d, what, dt, nchains * 5 * niter / dt / 1e6, float(np.max(Gelman_Rubin(sampled)))))
    x = np.concatenate([s[niter // 2:] for s in sampled[:64]])
    print("second-half sample mean of x0, x1: %.3f %.3f (truth 0, 0); var x0 %.1f (truth 100)" % (x[:, 0].mean(), x[:, 1].mean(), x[:, 0].var()))
