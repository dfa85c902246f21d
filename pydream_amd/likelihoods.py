"""Built-in likelihoods that run on the device ("batched device callback").

Each object is an ordinary Python callable ``f(x[d]) -> float`` (so it also works with the
reference), and carries a descriptor that ``run_dream`` hands to ``dz_set_likelihood_*`` so the
evaluation happens inside the HIP kernels instead of on the host.
"""
import numpy as np


class MVNormalLogLike:
    """log_F - 1/2 (x-mu)^T P (x-mu)  (pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52).

    factorize=True hands the device the upper-triangular factor U of P = U^T U (half the work of the
    dense form; agrees with it to ~1e-12); factorize=False uses P itself."""

    def __init__(self, precision, mu=None, log_F=0.0, factorize=True):
        P = np.asarray(precision, dtype=float)
        self.d = P.shape[0]
        self.precision = P
        self.mu = np.zeros(self.d) if mu is None else np.asarray(mu, dtype=float)
        self.log_F = float(log_F)
        self.factorize = bool(factorize)
        self.U = np.linalg.cholesky((P + P.T) / 2.0).T if self.factorize else None

    def __call__(self, x):
        v = np.asarray(x, dtype=float) - self.mu
        return self.log_F - .5 * np.sum(v * np.dot(self.precision, v))

    def _dz_apply(self, engine):
        if self.factorize:
            engine.set_likelihood_mvn(self.mu, self.U, 1, self.log_F)
        else:
            engine.set_likelihood_mvn(self.mu, self.precision, 0, self.log_F)


class GaussianMixtureLogLike:
    """log sum_j exp(-1/2 |x - mu_j|^2 + log_F_j)  (pydream/examples/mixturemodel/mixturemodel.py:37-48)."""

    def __init__(self, mu, log_F):
        self.mu = np.atleast_2d(np.asarray(mu, dtype=float))
        self.log_F = np.asarray(log_F, dtype=float)
        self.d = self.mu.shape[1]

    @classmethod
    def from_weights(cls, mu, weights):
        mu = np.atleast_2d(np.asarray(mu, dtype=float))
        d = mu.shape[1]
        return cls(mu, np.log(np.asarray(weights, dtype=float)) - (d / 2.) * np.log(2 * np.pi))

    def __call__(self, x):
        log_lh = -.5 * np.sum((np.asarray(x, dtype=float) - self.mu) ** 2, axis=1) + self.log_F
        m = np.max(log_lh)
        return np.log(np.sum(np.exp(log_lh - m))) + m

    def _dz_apply(self, engine):
        engine.set_likelihood_mixture(self.mu, self.log_F)


KERNEL_SIGNATURE = 'extern "C" __global__ void NAME(const double* X, long long n, int d, int ld, double* like, const void* data)'


def compile_device_kernel(source, out_path=None, extra_flags=(), arch="gfx950"):
    """HIP source text -> a code object (.hsaco) for DeviceKernelLogLike: `hipcc --offload-arch=<arch> --genco` (arch: the engine is built
    for gfx950, the MI355X; the argument is there so that an error names what was asked for).
    -ffp-contract=off is on by default so that a density written with + and * rounds like the same expression in numpy (a host
    likelihood and its device twin then make the same accept / reject decisions bit for bit); pass extra_flags=("-ffp-contract=fast",)
    to let the compiler fuse.  Without out_path the code object is CACHED under a key of (source, flags, arch) in
    $DREAMZS_KERNEL_CACHE (default ~/.cache/dreamzs_kernels, else the temporary directory): the same source is compiled once, by
    whichever process or unpickled copy asks first (advisor, round 5: a fresh temporary directory and a fresh hipcc run per call).
    Returns the path."""
    import hashlib
    import os
    import subprocess
    import tempfile
    text = source if "hip_runtime.h" in source else "#include <hip/hip_runtime.h>\n" + source
    flags = ["--offload-arch=%s" % arch, "--genco", "--no-gpu-bundle-output", "-O3", "-std=c++17", "-ffp-contract=off"] + list(extra_flags)
    cached = out_path is None
    if cached:
        key = hashlib.sha256(("\0".join(flags) + "\0" + text).encode()).hexdigest()[:32]
        cdir = os.environ.get("DREAMZS_KERNEL_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "dreamzs_kernels")
        try:
            os.makedirs(cdir, exist_ok=True)
            if not os.access(cdir, os.W_OK):
                raise OSError("not writable")
        except OSError:
            cdir = os.path.join(tempfile.gettempdir(), "dreamzs_kernels_%d" % os.getuid())
            os.makedirs(cdir, exist_ok=True)
        out_path = os.path.join(cdir, key + ".hsaco")
        if os.path.exists(out_path) and os.path.getsize(out_path) > 0:
            return out_path
    tmp = "%s.%d.tmp" % (out_path, os.getpid())          # (built beside the target, renamed into place: concurrent builders never see half a file)
    src = tmp + ".hip"
    with open(src, "w") as f:
        f.write(text)
    hipcc = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc") if c and os.path.exists(c)), "hipcc")
    try:
        res = subprocess.run([hipcc] + flags + ["-o", tmp, src], capture_output=True, text=True)
    except OSError as ex:          # (no compiler on this machine: say so -- a code object built elsewhere can be given by path)
        os.remove(src)
        raise Exception("hipcc failed for the device likelihood: cannot run %r (%s); set HIPCC, or build the code object where ROCm is installed and pass its path" % (hipcc, ex))
    try:
        os.remove(src) if cached else os.replace(src, out_path + ".hip")
    except OSError:
        pass
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise Exception("hipcc failed for the device likelihood (--offload-arch=%s):\n%s" % (arch, res.stderr[-4000:]))
    os.replace(tmp, out_path)
    return out_path


FUNCTION_SIGNATURE = '__device__ double NAME(const double* x, int d, const void* data, int lane)'

# The translation unit around a user's wave-level device function: the batch kernel the engine's multi-kernel path launches (one wave per
# point) and the persistent kernels -- csrc/dz_megakernel.h generations_wave_body, the kernel the built-in mixture runs in, with the user's
# function in the likelihood's place.  The names of the latter carry the layout generation of the structures they take (DZ_USER_ABI).
_FUNCTION_TU = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include "dz_device.h"
__device__ __forceinline__ double dz_wave_sum(double v) { return dz::wave_bfly(v); }      // the xor butterfly over the wave's 64 lanes: every lane gets the total
%(source)s
#define DZ_TEMPLATES_ONLY
#include "dz_kernels.h"
#undef DZ_TEMPLATES_ONLY
#include "dz_megakernel.h"
namespace dz {
struct UserLike {
    template <int NCH>
    DZ_DEV static void eval(const Params& p, const double* rows, int LDP, int n, int lane, double* lh, double* out)
    {
        for (int i = 0; i < n; ++i) {
            const double v = %(name)s(rows + (size_t)i * LDP, p.d, p.udata, lane);
            if (lane == 0) out[i] = nan_to_ninf(v);
        }
    }
};
}
#define DZ_USER_CAT2(a, b) a##b
#define DZ_USER_CAT(a, b) DZ_USER_CAT2(a, b)
extern "C" __global__ __launch_bounds__(1024) void DZ_USER_CAT(dz_user_generations_v, DZ_USER_ABI)(const dz::Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0,
                                                                                                     int64_t zappend, int seg0, dz::Publish pub)
{
    dz::generations_wave_body<false, false, dz::UserLike>(pp, g0, ngen, M0, trace_slot0, zappend, seg0, pub);
}
extern "C" __global__ __launch_bounds__(1024) void DZ_USER_CAT(dz_user_generations_full_v, DZ_USER_ABI)(const dz::Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0,
                                                                                                          int64_t zappend, int seg0, dz::Publish pub)
{
    dz::generations_wave_body<true, false, dz::UserLike>(pp, g0, ngen, M0, trace_slot0, zappend, seg0, pub);
}
// ... and for 128 < d <= 256 (a lane owns four dimensions of the chain's state)
extern "C" __global__ __launch_bounds__(1024) void DZ_USER_CAT(dz_user_generations_wide_v, DZ_USER_ABI)(const dz::Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0,
                                                                                                          int64_t zappend, int seg0, dz::Publish pub)
{
    dz::generations_wave_body<false, false, dz::UserLike, 2>(pp, g0, ngen, M0, trace_slot0, zappend, seg0, pub);
}
extern "C" __global__ __launch_bounds__(1024) void DZ_USER_CAT(dz_user_generations_wide_full_v, DZ_USER_ABI)(const dz::Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0,
                                                                                                               int64_t zappend, int seg0, dz::Publish pub)
{
    dz::generations_wave_body<true, false, dz::UserLike, 2>(pp, g0, ngen, M0, trace_slot0, zappend, seg0, pub);
}
extern "C" __global__ __launch_bounds__(256) void dz_user_batch(const double* X, long long n, int d, int ld, double* like, const void* data)
{
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const double v = %(name)s(X + (size_t)i * ld, d, data, lane);
    if (lane == 0) like[i] = v;
}
"""


def csrc_dir():
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


class DeviceFunctionLogLike:
    """A user-written HIP DEVICE FUNCTION as the likelihood, evaluated by one wave per point -- and INSIDE the persistent generation kernel:

        SRC = '''
        __device__ double sphere(const double* x, int d, const void* data, int lane) {     // all 64 lanes call it; x: the point (d doubles)
            const double* c = (const double*)data;
            double acc = 0.0;
            for (int j = lane; j < d; j += 64) { const double t = x[j] - c[j]; acc += t * t; }
            return -0.5 * dz_wave_sum(acc);                                                // the same value in every lane
        }'''
        like = DeviceFunctionLogLike(SRC, "sphere", ndim=d, data=centre, always_finite=True)
        sampled, log_ps = run_dream(parameters, like, nchains=4096, multitry=5, ...)

    The function's signature is FUNCTION_SIGNATURE; x may point to LDS or global memory (d doubles; what follows them is not the
    function's to read); `dz_wave_sum(double)` (the xor butterfly over the wave's 64 lanes, every lane gets the total) and the helpers of
    csrc/dz_device.h (namespace dz) are in scope.  The source is compiled once (hipcc on first use, cached like compile_device_kernel's
    objects; the key includes the engine's own headers) into a code object with (a) a batch kernel for the engine's multi-kernel path and
    (b) the persistent kernels of csrc/dz_megakernel.h with this function in the place of the built-in mixture -- the same launches, the
    same bits as (a), at the persistent kernels' rate instead of the multi-kernel path's.  They run where the mixture's would: d <= 256,
    multitry 1 or 3..32 tries, and always_finite=True (a density that may be -inf for a whole proposal set needs the multi-kernel path's
    redraw rounds, Dream.py:281-289).  host: an optional Python twin f(x[d]) -> float for calls on the host (Model.total_logp)."""

    def __init__(self, source, name, ndim, data=None, always_finite=False, host=None, extra_flags=(), path=None):
        """path: a code object built beforehand from this source (`DeviceFunctionLogLike(...).code_object()` on a machine with hipcc and THIS
        version of the package: the persistent kernels' names carry the layout generation, a stale object just runs the multi-kernel path)"""
        if source is None and path is None:
            raise ValueError("give the device function's HIP source (or the path of a code object built from it)")
        self.name, self.d, self.source = name, int(ndim), source
        self.data = None if data is None else np.ascontiguousarray(data)
        self.always_finite, self.host, self.extra_flags = bool(always_finite), host, tuple(extra_flags)
        self.path = path
        self._eval_engine = None

    def code_object(self):
        if self.path is None:
            import glob
            import hashlib
            import os
            hdr = hashlib.sha256()
            for h in sorted(glob.glob(os.path.join(csrc_dir(), "*.h"))):       # (the object is only as current as the headers it was built against)
                with open(h, "rb") as fh:
                    hdr.update(fh.read())
            text = _FUNCTION_TU % dict(source=self.source, name=self.name) + "\n// headers " + hdr.hexdigest() + "\n"
            self.path = compile_device_kernel(text, extra_flags=("-I" + csrc_dir(), "-Wno-unused-value", "-Wno-unused-result") + self.extra_flags)
        return self.path

    def _dz_apply(self, engine):
        engine.set_likelihood_module(self.code_object(), "dz_user_batch", 64, self.data, self.always_finite)

    def __call__(self, x):
        if self.host is not None:
            return self.host(np.asarray(x, dtype=float))
        if self._eval_engine is None:
            from . import _capi
            self._eval_engine = _capi.Engine(nchains=3, ndim=self.d, history_capacity=8)
            self._dz_apply(self._eval_engine)
        return float(self._eval_engine.eval_logp(np.asarray(x, dtype=float).reshape(1, self.d))[1][0])

    def __getstate__(self):
        st = dict(self.__dict__); st["_eval_engine"] = None
        return st


class DeviceKernelLogLike:
    """A user-written HIP kernel as the likelihood -- the batched device callback for ANY model (the reference accepts any callable,
    pydream/model.py:17-32; a Python callable runs through the host callback at ~40 k proposals/s, this runs where the built-in
    densities' kernels run).

        like = DeviceKernelLogLike(source=SRC, name="my_logp", ndim=d, data=np.array([...]))     # compiled with hipcc on first use
        like = DeviceKernelLogLike(path="model.hsaco", name="my_logp", ndim=d)                   # or a code object built beforehand
        sampled, log_ps = run_dream(parameters, like, nchains=4096, ...)

    The kernel's signature is KERNEL_SIGNATURE: point i is the row X + i * ld, like[i] receives its log likelihood (-inf allowed); `data`
    is the device copy of the `data` array.  lanes_per_point=1: one thread per point; 64: one wave per point (coalesced row reads, the
    kernel reduces over its lanes itself).  always_finite=True promises the density is finite wherever the priors are (skips the
    per-generation "every try impossible?" check of Dream.py:281-289).  host: an optional Python twin f(x[d]) -> float used when the
    object is called on the host (Model.total_logp); without it a call evaluates the point on the device."""

    def __init__(self, name, ndim, source=None, path=None, data=None, lanes_per_point=1, always_finite=False, host=None, extra_flags=()):
        if (source is None) == (path is None):
            raise ValueError("give either the kernel's HIP source or the path of a gfx950 code object")
        self.name, self.d, self.source, self.path = name, int(ndim), source, path
        self.data = None if data is None else np.ascontiguousarray(data)
        self.lanes_per_point, self.always_finite, self.host, self.extra_flags = int(lanes_per_point), bool(always_finite), host, tuple(extra_flags)
        self._eval_engine = None

    def code_object(self):
        if self.path is None:
            self.path = compile_device_kernel(self.source, extra_flags=self.extra_flags)
        return self.path

    def _dz_apply(self, engine):
        engine.set_likelihood_module(self.code_object(), self.name, self.lanes_per_point, self.data, self.always_finite)

    def __call__(self, x):
        if self.host is not None:
            return self.host(np.asarray(x, dtype=float))
        if self._eval_engine is None:
            from . import _capi
            self._eval_engine = _capi.Engine(nchains=3, ndim=self.d, history_capacity=8)
            self._dz_apply(self._eval_engine)
        return float(self._eval_engine.eval_logp(np.asarray(x, dtype=float).reshape(1, self.d))[1][0])

    def __getstate__(self):                      # (the evaluation engine is a device handle: not part of the object's value)
        st = dict(self.__dict__); st["_eval_engine"] = None
        return st
