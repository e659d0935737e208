"""ctypes loader of the in-tree HIP library (gymgo_amd/libgymgo_amd.so) + tensor plumbing.

There is NO CPU fallback: if the shared library is missing, or a tensor is not a contiguous
uint8/int32 ROCm device tensor, the call raises.  PyTorch is used only for device memory and the
current HIP stream; every entry point of include/gymgo_amd.h is bound here with plain pointers.
"""
import ctypes
import os
import threading

import torch  # must be imported before the library so both share ONE HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgymgo_amd.so')   # always the in-tree build: no override, no search path
ABI_VERSION = 5                                      # GG_ABI_VERSION of include/gymgo_amd.h

EXPORTS = (
    'gg_version', 'gg_device_cus', 'gg_batch_next_states', 'gg_batch_next_states_ws', 'gg_batch_invalid_mask', 'gg_batch_areas',
    'gg_batch_children', 'gg_batch_children_offsets', 'gg_batch_children_compact', 'gg_batch_rollout', 'gg_batch_env_step', 'gg_batch_sample_actions', 'gg_batch_update_pieces', 'gg_batch_reset_finished', 'gg_packed_words', 'gg_batch_pack_states',
    'gg_batch_unpack_states', 'gg_batch_next_states_packed', 'gg_batch_rollout_packed', 'gg_batch_env_step_packed',
    'gg_batch_children_packed', 'gg_batch_play_moves', 'gg_batch_play_moves_packed', 'gg_tracked_words', 'gg_batch_track_states',
    'gg_batch_untrack_states', 'gg_batch_rollout_tracked', 'gg_batch_play_moves_tracked', 'gg_batch_env_step_tracked',
    'gg_rng_seed', 'gg_batch_env_step_tracked_weighted', 'gg_batch_sample_weighted', 'gg_batch_sample_weighted_rows',
    'gg_batch_symmetry', 'gg_batch_symmetry_rows', 'gg_batch_env_step_scored',
)

_vp, _i64, _i32, _u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64
_SIGNATURES = {
    'gg_version': ([], _i32),
    'gg_device_cus': ([], _i32),
    'gg_batch_next_states': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_next_states_ws': ([_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_invalid_mask': ([_vp, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_areas': ([_vp, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_children': ([_vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_children_offsets': ([_vp, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_children_compact': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_rollout': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
    'gg_batch_env_step': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _i32, _i32, _vp], _i32),
    'gg_batch_env_step_scored': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _i32, _i32, _vp], _i32),
    'gg_batch_sample_actions': ([_vp, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_update_pieces': ([_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_reset_finished': ([_vp, _i64, _i32, _vp], _i32),
    'gg_packed_words': ([_i32], _i32),
    'gg_batch_pack_states': ([_vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_unpack_states': ([_vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_next_states_packed': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_rollout_packed': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
    'gg_batch_env_step_packed': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _i32, _i32, _vp], _i32),
    'gg_batch_children_packed': ([_vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_play_moves': ([_vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_play_moves_packed': ([_vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_tracked_words': ([_i32], _i32),
    'gg_batch_track_states': ([_vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_untrack_states': ([_vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_rollout_tracked': ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
    'gg_batch_play_moves_tracked': ([_vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_env_step_tracked': ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _i32, _i32, _vp], _i32),
    'gg_rng_seed': ([_vp, _u64, _i64, _i64, _vp], _i32),
    'gg_batch_env_step_tracked_weighted': ([_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _i32, _i32, _vp], _i32),
    'gg_batch_sample_weighted': ([_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_sample_weighted_rows': ([_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _vp], _i32),
    'gg_batch_symmetry': ([_vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    'gg_batch_symmetry_rows': ([_vp, _i32, _vp, _vp, _i64, _i32, _vp], _i32),
}

_lib = None


class GymGoNativeError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    csrc = os.path.join(_HERE, 'csrc')
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.h'))]
    srcs.append(os.path.join(os.path.dirname(_HERE), 'include', 'gymgo_amd.h'))
    stale = not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(['make', '-C', os.path.join(_HERE, 'csrc'), '-s', '-B', '-j3'])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GymGoNativeError(
                'HIP library %s not built (run `make -C gymgo_amd/csrc` or __graft_entry__.build()); '
                'gymgo_amd has no CPU fallback' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export what the header declares
            fn.argtypes, fn.restype = argtypes, restype
        if L.gg_version() != ABI_VERSION:
            raise GymGoNativeError('%s has ABI version %d, this package binds version %d: rebuild it '
                                   '(make -C gymgo_amd/csrc)' % (LIB_PATH, L.gg_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise GymGoNativeError('%s failed with code %d (%s)' % (
            what, code, 'bad argument' if code < 0 else 'hipError_t'))


_stream_override = threading.local()   # .raw: a hipStream_t that replaces torch's current stream (GoVecEnvParts)


def stream_ptr(device=None):
    """torch's current stream on `device` as a hipStream_t: launches go where the caller's torch work goes.  (A
    GoVecEnvParts sub-batch steps on its own stream: it names that stream here for the duration of its call instead of
    switching torch's current stream, which costs more host time than the launch.)"""
    raw = getattr(_stream_override, 'raw', None)
    if raw is not None:
        dev = getattr(_stream_override, 'device', None)     # the override names a stream of ONE device
        if dev is None or _device_index(device) == dev:
            return raw
    return current_raw_stream(device)


def _device_index(device):
    if isinstance(device, str):
        device = torch.device(device)
    idx = device.index if isinstance(device, torch.device) else device
    return torch.cuda.current_device() if idx is None else idx


def current_raw_stream(device=None):
    """hipStream_t of torch's current stream on `device` (an int; 0 = the null stream)."""
    try:
        if isinstance(device, str):           # 'cuda' / 'cuda:1', as torch's own factory functions accept
            device = torch.device(device)
        idx = device.index if isinstance(device, torch.device) else device
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)
    except (AttributeError, TypeError):   # an older / newer torch without the raw getter
        return torch.cuda.current_stream(device).cuda_stream


class stream_override:
    """with stream_override(raw, device): every launch of this thread ON `device` goes to the hipStream_t `raw` (launches on
    another device keep torch's current stream there; device None = whatever device the launch is on)."""

    def __init__(self, raw, device=None):
        self.raw = raw
        self.device = None if device is None else _device_index(device)

    def __enter__(self):
        self.prev = (getattr(_stream_override, 'raw', None), getattr(_stream_override, 'device', None))
        _stream_override.raw, _stream_override.device = self.raw, self.device

    def __exit__(self, *exc):
        _stream_override.raw, _stream_override.device = self.prev


_hip = None


def hip_runtime():
    """The HIP runtime torch and the library already share, bound for the three stream-ordering calls GoVecEnvParts
    makes per step (torch's Stream.wait_stream creates and destroys an event per call: 8 us of host time against 2) and the
    stream wait of GoEnv.step."""
    global _hip
    if _hip is None:
        # The runtime ALREADY in the process (torch's and the library's): its path is read from /proc/self/maps and that exact
        # file is opened - dlopen by soname could bring in a second runtime when a torch wheel bundles its own copy under
        # another name, and events created in one runtime mean nothing to streams of the other.
        H, paths = None, []
        try:
            with open('/proc/self/maps') as f:
                for line in f:
                    # (the path is everything from the first '/' on: it may hold spaces; a '(deleted)' suffix cannot be opened)
                    path = line[line.index('/'):].rstrip('\n') if '/' in line else ''
                    if path.endswith(' (deleted)'):
                        continue
                    if 'libamdhip64' in os.path.basename(path) and path not in paths:
                        paths.append(path)
        except OSError:
            pass
        for name in paths + ['libamdhip64.so']:
            try:
                H = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if H is None:
            raise GymGoNativeError('libamdhip64.so not found (it is loaded with torch)')
        H.hipEventCreateWithFlags.argtypes, H.hipEventCreateWithFlags.restype = [ctypes.POINTER(_vp), ctypes.c_uint], _i32
        H.hipEventRecord.argtypes, H.hipEventRecord.restype = [_vp, _vp], _i32
        H.hipStreamWaitEvent.argtypes, H.hipStreamWaitEvent.restype = [_vp, _vp, ctypes.c_uint], _i32
        H.hipEventDestroy.argtypes, H.hipEventDestroy.restype = [_vp], _i32
        H.hipStreamQuery.argtypes, H.hipStreamQuery.restype = [_vp], _i32
        H.hipStreamSynchronize.argtypes, H.hipStreamSynchronize.restype = [_vp], _i32
        H.hipStreamIsCapturing.argtypes, H.hipStreamIsCapturing.restype = [_vp, ctypes.POINTER(ctypes.c_int)], _i32
        if torch.cuda.is_available():
            # the same runtime as torch's: a torch stream must be a stream it knows.  Probed with hipStreamIsCapturing,
            # which is legal on a capturing stream (hipStreamQuery would invalidate the capture) and reports an unknown
            # handle (400 hipErrorInvalidHandle / 709 hipErrorContextIsDestroyed) - any other code (a sticky asynchronous
            # error of earlier work, say) is not a statement about the runtime and is left to the call that caused it
            cap = ctypes.c_int(0)
            rc = H.hipStreamIsCapturing(current_raw_stream(None), ctypes.byref(cap))
            if rc in (400, 709):
                raise GymGoNativeError('the HIP runtime opened for stream ordering (%s) does not know torch\'s stream '
                                       '(hipStreamIsCapturing -> %d): two runtimes in one process?' % (getattr(H, '_name', '?'), rc))
        _hip = H
    return _hip


def hip_event():
    """A HIP event without timing (usable inside a stream capture)."""
    ev = _vp()
    check(hip_runtime().hipEventCreateWithFlags(ctypes.byref(ev), 2), 'hipEventCreateWithFlags')   # hipEventDisableTiming
    return ev


def dev_ptr(t, dtype, name):
    """Pointer of a contiguous device tensor of `dtype`; raises instead of silently copying."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GymGoNativeError('%s must be a ROCm device tensor (got %r); gymgo_amd has no CPU path'
                               % (name, type(t) if not isinstance(t, torch.Tensor) else t.device))
    if t.dtype != dtype or not t.is_contiguous():
        raise GymGoNativeError('%s must be contiguous %s (got %s, contiguous=%s)'
                               % (name, dtype, t.dtype, t.is_contiguous()))
    return t.data_ptr()
