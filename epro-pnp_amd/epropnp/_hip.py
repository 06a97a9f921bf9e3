"""ctypes binding of libepropnp_hip.so (C ABI declared in include/epropnp_hip.h).

There is no CPU implementation behind this module: if the shared library has not been built
(`python epro-pnp_amd/build.py`) or a tensor does not live on a HIP device, calls raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPROPNP_LIB: alternative build of the SAME HIP library (kernel-tuning variants, tools/tune.py); otherwise the in-tree
# build (epro-pnp_amd/lib, `python epro-pnp_amd/build.py`) or the copy a `pip install .` places inside the package
_IN_TREE = os.path.join(os.path.dirname(_HERE), 'lib', 'libepropnp_hip.so')
_INSTALLED = os.path.join(_HERE, '_lib', 'libepropnp_hip.so')
LIB_PATH = os.environ.get('EPROPNP_LIB') or (_INSTALLED if (os.path.exists(_INSTALLED) and not os.path.exists(_IN_TREE)) else _IN_TREE)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def tune(key):
    """EPROPNP_TUNE="key=value;key2;...": the ONE string behind which launch-shape overrides, implementation selectors and
    phase ablation live (csrc/pnp_host.h: tune_value; tools/tune.py and the shape tests set it, users do not).  Returns the text
    behind `key=` ('' for a bare key) or None.  Read on every call: tests change the variable at run time."""
    text = os.environ.get('EPROPNP_TUNE')
    if not text:
        return None
    for item in text.split(';'):
        k, _, v = item.partition('=')
        if k == key:
            return v
    return None


class Problem(C.Structure):
    _fields_ = [('x3d', C.c_void_p), ('x2d', C.c_void_p), ('w2d', C.c_void_p), ('cam_mats', C.c_void_p),
                ('lb', C.c_void_p), ('ub', C.c_void_p), ('delta', C.c_void_p), ('z_min', C.c_float),
                ('num_obj', C.c_int32), ('num_pts', C.c_int32), ('dof', C.c_int32),
                ('huber_eps', C.c_float), ('status', C.c_void_p), ('delta_stats', C.c_void_p),
                ('delta_relative', C.c_float)]


class LmParams(C.Structure):
    _fields_ = [('num_iter', C.c_int32), ('fast_mode', C.c_int32), ('min_lm_diagonal', C.c_float),
                ('max_lm_diagonal', C.c_float), ('min_relative_decrease', C.c_float),
                ('initial_trust_region_radius', C.c_float), ('max_trust_region_radius', C.c_float),
                ('eps', C.c_float)]


class AmisParams(C.Structure):
    _fields_ = [('mc_samples', C.c_int32), ('num_iter', C.c_int32), ('eps', C.c_float),
                ('acg_mle_iter', C.c_int32), ('acg_dispersion', C.c_float), ('seed', C.c_uint64),
                ('offset', C.c_uint64), ('offset_dev', C.c_void_p), ('split_scratch', C.c_void_p),
                ('split_scratch_bytes', C.c_uint64), ('advance', C.c_void_p), ('advance_ticket', C.c_void_p),
                ('advance_count', C.c_int32)]


class McParams(C.Structure):
    _fields_ = [('lm', LmParams), ('amis', AmisParams), ('normalize', C.c_int32), ('init_mode', C.c_int32),
                ('rslm_lm', LmParams), ('rslm_points', C.c_int32), ('rslm_proposals', C.c_int32),
                ('rslm_seed', C.c_uint64), ('rslm_offset', C.c_uint64), ('rslm_offset_dev', C.c_void_p),
                ('rslm_inds', C.c_void_p), ('rslm_rot', C.c_void_p), ('rslm_scratch', C.c_void_p),
                ('rslm_scratch_bytes', C.c_uint64), ('lm_scratch', C.c_void_p), ('lm_scratch_bytes', C.c_uint64)]


ABI_VERSION = 6
_lib = None


def _declare(lib):
    vp, i32 = C.c_void_p, C.c_int32
    lib.epropnp_abi_version.restype = C.c_int
    lib.epropnp_last_error.restype = C.c_char_p
    lib.epropnp_noise_stride.argtypes = [C.c_int]
    lib.epropnp_profile_enable.argtypes = [C.c_int]
    lib.epropnp_amis_forward_split_bytes.argtypes = [C.POINTER(Problem), C.c_int32, C.c_int32]
    lib.epropnp_amis_forward_split_bytes.restype = C.c_uint64
    lib.epropnp_async_status.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.epropnp_async_status.restype = C.c_int
    lib.epropnp_async_status_word.restype = C.POINTER(C.c_int32)
    lib.epropnp_profile_read.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.epropnp_monte_carlo_forward.argtypes = [C.POINTER(Problem), C.POINTER(McParams)] + [vp] * 16
    lib.epropnp_evaluate_cost.argtypes = [C.POINTER(Problem), vp, i32, vp, vp]
    lib.epropnp_normal_equations.argtypes = [C.POINTER(Problem), vp, i32, vp, vp, vp, vp]
    lib.epropnp_cost_pose_cam_grad.argtypes = [C.POINTER(Problem), vp, vp, i32, i32, vp, vp, vp]
    lib.epropnp_lm_solve.argtypes = [C.POINTER(Problem), C.POINTER(LmParams), vp, vp, vp, vp, vp, vp, C.c_uint64, vp]
    lib.epropnp_lm_solve_split_bytes.argtypes = [C.POINTER(Problem), C.POINTER(LmParams)]
    lib.epropnp_lm_solve_split_bytes.restype = C.c_uint64
    lib.epropnp_amis_forward.argtypes = [C.POINTER(Problem), C.POINTER(AmisParams), vp, vp, vp, vp, vp, vp, vp]
    lib.epropnp_amis_backward.argtypes = [C.POINTER(Problem), vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.epropnp_amis_backward_split.argtypes = [C.POINTER(Problem), vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.epropnp_gn_step_forward.argtypes = [C.POINTER(Problem), C.c_float, vp, vp, vp]
    lib.epropnp_gn_step_backward.argtypes = [C.POINTER(Problem), C.c_float, vp, vp, vp, vp, vp, vp, vp]
    lib.epropnp_pose_opt_plus_forward.argtypes = [C.POINTER(Problem), C.c_float, vp, vp, vp]
    lib.epropnp_pose_opt_plus_backward.argtypes = [C.POINTER(Problem), C.c_float, vp, vp, vp, vp, vp, vp, vp]
    lib.epropnp_rslm_draw.argtypes = [vp, i32, i32, i32, i32, C.c_uint64, C.c_uint64, vp, vp]
    lib.epropnp_center_points.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.epropnp_shift_poses.argtypes = [vp, vp, i32, i32, i32, C.c_float, vp, vp]
    lib.epropnp_shift_poses_backward.argtypes = [vp, vp, vp, i32, i32, i32, C.c_float, vp, vp]
    lib.epropnp_prepare_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.epropnp_prepare_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.epropnp_prepare_dense_forward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.epropnp_prepare_dense_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.epropnp_rslm_solve.argtypes = [C.POINTER(Problem), C.POINTER(LmParams), i32, i32, C.c_uint64, C.c_uint64, vp, vp, vp,
                                       vp, vp, vp, C.c_uint64, vp]
    lib.epropnp_rslm_solve_scratch_bytes.argtypes = [C.POINTER(Problem), i32]
    lib.epropnp_rslm_solve_scratch_bytes.restype = C.c_uint64
    lib.epropnp_adaptive_delta.argtypes = [vp, vp, i32, i32, C.c_float, vp, vp, vp]
    lib.epropnp_mc_loss_forward.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.epropnp_mc_loss_backward.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.epropnp_mc_loss_reduce.argtypes = [vp, vp, i32, C.c_float, C.c_float, vp, i32, C.c_int64, vp, vp, vp]
    lib.epropnp_exchange_pack.argtypes = [vp, C.c_uint64, vp, i32, vp, C.c_uint64, C.c_float, vp, i32, vp, vp]
    lib.epropnp_mc_loss_reduce_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    for name in ('evaluate_cost', 'normal_equations', 'lm_solve', 'amis_forward', 'amis_backward', 'adaptive_delta',
                 'mc_loss_forward', 'mc_loss_backward', 'rslm_draw', 'gn_step_forward', 'gn_step_backward', 'rslm_solve', 'center_points', 'shift_poses', 'prepare_forward', 'prepare_backward', 'pose_opt_plus_forward', 'pose_opt_plus_backward', 'shift_poses_backward', 'prepare_dense_forward',
                 'prepare_dense_backward', 'amis_backward_split', 'monte_carlo_forward', 'cost_pose_cam_grad', 'mc_loss_reduce',
                 'mc_loss_reduce_backward'):
        getattr(lib, 'epropnp_' + name).restype = C.c_int
    return lib


EXPORTS = ('epropnp_abi_version', 'epropnp_last_error', 'epropnp_noise_stride', 'epropnp_profile_enable',
           'epropnp_profile_reset', 'epropnp_profile_read', 'epropnp_evaluate_cost',
           'epropnp_normal_equations', 'epropnp_lm_solve', 'epropnp_amis_forward', 'epropnp_amis_backward',
           'epropnp_adaptive_delta', 'epropnp_mc_loss_forward', 'epropnp_mc_loss_backward', 'epropnp_rslm_draw',
           'epropnp_gn_step_forward', 'epropnp_gn_step_backward', 'epropnp_rslm_solve',
           'epropnp_center_points', 'epropnp_shift_poses', 'epropnp_prepare_forward', 'epropnp_prepare_backward',
           'epropnp_pose_opt_plus_forward', 'epropnp_pose_opt_plus_backward', 'epropnp_shift_poses_backward',
           'epropnp_prepare_dense_forward', 'epropnp_prepare_dense_backward', 'epropnp_amis_backward_split',
           'epropnp_monte_carlo_forward', 'epropnp_cost_pose_cam_grad', 'epropnp_async_status',
           'epropnp_async_status_word', 'epropnp_amis_forward_split_bytes', 'epropnp_rslm_solve_scratch_bytes',
           'epropnp_lm_solve_split_bytes', 'epropnp_mc_loss_reduce', 'epropnp_mc_loss_reduce_backward', 'epropnp_exchange_pack')


def lib():
    """The loaded library; raises if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP extension has not been built '
                '(run `python epro-pnp_amd/build.py`); there is no CPU fallback')
        loaded = _declare(C.CDLL(LIB_PATH))
        if loaded.epropnp_abi_version() != ABI_VERSION:
            raise RuntimeError('libepropnp_hip.so ABI version mismatch')
        _lib = loaded
    return _lib


_torch_ext = False      # not tried yet


def torch_ext():
    """The C++ autograd nodes (lib/_epropnp_torch.so, csrc/torch_binding.cpp: same C ABI, no interpreter in the backward)
    or None when the module was not built / EPROPNP_NO_TORCH_EXT is set / another library build is selected with
    EPROPNP_LIB (the module is linked to the default one).  Either way the kernels are the library's."""
    global _torch_ext
    if _torch_ext is False:
        _torch_ext = None
        path = os.path.join(os.path.dirname(LIB_PATH), '_epropnp_torch.so')
        if os.path.exists(path) and not os.environ.get('EPROPNP_NO_TORCH_EXT') and not os.environ.get('EPROPNP_LIB'):
            import importlib.util
            lib()                        # the HIP library first: a missing / mismatching build fails with a clear message
            try:
                spec = importlib.util.spec_from_file_location('_epropnp_torch', path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.abi_version() == ABI_VERSION and mod.mc_params_size() == C.sizeof(McParams):
                    _torch_ext = mod
            except (ImportError, OSError) as e:      # built against another torch: same kernels through the ctypes nodes
                import warnings
                warnings.warn(f'{path} could not be loaded ({e}); using the ctypes binding of the same library')
    return _torch_ext


def check_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on a HIP device (got {t.device}); the EPro-PnP HIP path has no CPU fallback')


def on_hip_path(*tensors):
    """True when the fused kernels can serve these tensors: fp32 on a HIP device."""
    return all(t.dtype == torch.float32 and t.is_cuda for t in tensors)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)     # hipStream_t as an int, without a Stream object


def stream_of(t):
    idx, cur = t.device.index, torch.cuda.current_device()
    if idx is not None and idx != cur:
        # kernels are launched on the calling thread's current device: a foreign stream would fail inside HIP
        raise RuntimeError(f'epropnp: tensors live on {t.device} but the current device is cuda:{cur}; call '
                           f'torch.cuda.set_device({idx}) (one process per GPU, as under torch.distributed)')
    if _raw_stream is not None:
        return _raw_stream(cur)
    return torch.cuda.current_stream(t.device).cuda_stream


def profile(enable=None, reset=False):
    """Per-stage HIP-event timing inside the library (epropnp_profile_*): switch it on / off, optionally dropping what was
    recorded so far."""
    if reset:
        lib().epropnp_profile_reset()
    if enable is not None:
        lib().epropnp_profile_enable(int(bool(enable)))


def profile_read(stage):
    """(mean ms per launch, number of launches) of `stage` since the last reset; synchronises on the recorded events."""
    ms, n = C.c_float(0), C.c_int32(0)
    lib().epropnp_profile_read(stage.encode(), C.byref(ms), C.byref(n))
    return (ms.value if n.value else float('nan')), n.value


ST_LM_NOT_SPD, ST_NONFINITE_POSE = 1, 2          # include/epropnp_hip.h: the two events the reference raises on
ST_SPLIT_TIMEOUT = 16                            # a split kernel recomputed a sibling workgroup's share: slower, results unaffected
_status_words = {}


STATUS_MODE = os.environ.get('EPROPNP_ASYNC_STATUS', 'warn')     # 'warn' (default) | 'raise' | '0' (no status word at all)


_has_gpu = None


def _gpu_present():
    """torch.cuda.is_available(), asked once: it is a driver query (hipGetDeviceCount, ~50 us on an MI355X box) and every entry
    into the package passes through poll_status()."""
    global _has_gpu
    if _has_gpu is None:
        _has_gpu = bool(torch.cuda.is_available())
    return _has_gpu


def poll_status():
    """The asynchronous half of the error convention.  Kernels report a damped normal-equation system without a Cholesky
    factor / a non-finite pose into the library's host-mapped status word of the current device; reading it is a plain
    host load, so every entry into the package polls it -- at the first call after the failing kernel has run, without
    ever synchronising (`flush_status()` synchronises and polls).
    What happens then follows the reference's behaviour rather than its letter: `torch.linalg.solve` / `torch.inverse`
    (levenberg_marquardt.py:15-19,178-181) raise only on an exactly zero LU pivot, which the LM damping rules out, and
    otherwise hand NaN poses on silently -- the callers' losses zero NaN objects (monte_carlo_pose_loss.py:31).  So the
    default is a RuntimeWarning naming the object; EPROPNP_ASYNC_STATUS=raise (or `_hip.STATUS_MODE = 'raise'`) turns it
    into the RuntimeError, `with numerics_check():` does so for a block, synchronously.  Events the reference takes in its
    stride (Cholesky fallback of a proposal, non-finite log-weight) are dropped here."""
    dev = torch.cuda.current_device() if _gpu_present() else 0
    w = _status_words.get((id(_lib), dev))
    if w is None:
        w = lib().epropnp_async_status_word()
        _status_words[(id(_lib), dev)] = w if w else False
    if not w:
        return
    flags = w[0]
    if flags:
        first = w[1]
        w[0], w[1] = 0, 2 ** 31 - 1
        if flags & ST_SPLIT_TIMEOUT:
            _warn_split_degraded(first, dev)
        msg = None
        if flags & ST_LM_NOT_SPD:
            msg = (f'linalg.solve: the damped normal equations of object {first} are singular or not finite '
                   f'(reported asynchronously by an earlier EPro-PnP launch on device {dev})')
        elif flags & ST_NONFINITE_POSE:
            msg = (f'the solver produced a non-finite pose for object {first} (reported asynchronously by an earlier '
                   f'EPro-PnP launch on device {dev})')
        if msg is not None:
            if STATUS_MODE == 'raise':
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)


_split_warned = False


def _warn_split_degraded(first, dev):
    """EPROPNP_ST_SPLIT_TIMEOUT is a PERFORMANCE event: in a launch that splits an object over several workgroups (AMIS
    forward at <= 64 objects, LM solve beyond 2048 points per object) a workgroup did not see a sibling's partial sums in
    time -- the siblings were not all resident (CU mask, partitioned GPU, another kernel holding CUs) -- and recomputed them
    itself: same bits, more time.  Said once per process."""
    global _split_warned
    if not _split_warned:
        _split_warned = True
        import warnings
        warnings.warn(f'EPro-PnP: a workgroup-split launch on device {dev} (object {first}) found a sibling workgroup missing '
                      f'and recomputed its share -- results are unaffected, the launch was slower.  If this GPU is shared or '
                      f'partitioned, EPROPNP_FWD_SPLIT=1 / EPROPNP_LM_SPLIT=1 switch the AMIS forward / LM solve splits off.',
                      RuntimeWarning, stacklevel=4)


def flush_status():
    """Synchronise the current device and raise any pending numerical event now."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    poll_status()


def call(fn_name, *args):
    rc = getattr(lib(), fn_name)(*args)
    if rc != 0:
        raise RuntimeError(f'{fn_name} failed ({rc}): {lib().epropnp_last_error().decode()}')
