"""ctypes binding of libre2e_hip.so (C ABI declared in include/re2e.h).

The product path has NO fallback: if the shared library is missing or the device is not gfx950
every op raises.  ``load()`` only dlopen()s the library and checks the exported symbols (usable
on a CPU-only box for the build/ABI checks); the first kernel call needs a GPU.
"""
import atexit
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint, c_ulonglong, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RE2E_LIB selects another build of the same C ABI (A/B measurements of kernel changes inside one GPU session)
LIB_PATH = os.environ.get('RE2E_LIB') or os.path.join(_HERE, 'libre2e_hip.so')
ABI_VERSION = 319      # include/re2e.h RE2E_ABI_VERSION this table was written for (checked against the library in load())

ACT_NONE, ACT_TANH, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_SIGMOID_MASK_MUL = range(6)
LOSS_L2, LOSS_L1, LOSS_SMOOTH_L1, LOSS_BCE = range(4)
EUNSUPPORTED = -2      # RE2E_EUNSUPPORTED
STREAM_DEFAULT, STREAM_FILLER = 0, 1

P, I, L, F, Z = c_void_p, c_int, c_long, c_float, c_size_t

# name -> (restype, argtypes).  Must list every symbol of include/re2e.h (tests/test_abi.py).
SIGNATURES = {
    're2e_version': (I, []),
    're2e_warmup': (I, []),
    're2e_last_error': (c_char_p, []),
    're2e_device_ok': (I, []),
    're2e_stream_role': (I, [P, I]),
    're2e_gemm_workspace_bytes': (Z, [I, I, I, I, I]),
    're2e_gemm': (I, [I, I, I, I, I, P, L, P, L, P, L, P, P, I, F, P, P, P, I, P, Z, P]),
    're2e_gemm_nt_rows': (I, [I, I, I, P, L, P, L, P, L, P, P, I, F, P, I, I, P, Z, P]),
    're2e_gemm_tn_rows': (I, [I, I, I, P, L, P, L, P, L, F, P, I, I, P, Z, P]),
    're2e_fill_rows': (I, [P, L, I, P, I, F, P, I, P]),
    're2e_conv_igemm': (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, P, I, I, I, I, I, I, P, I, F, P]),
    're2e_conv3x3_relu_pool': (I, [P, I, I, I, I, P, I, P, P, P, P]),
    're2e_conv_igemm_masked': (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, P, I, I, I, I, I, I, P, P]),
    're2e_conv3x3_wino_workspace_bytes': (Z, [I, I]),
    're2e_conv3x3_wino': (I, [P, I, I, I, I, P, I, I, P, I, P, P, P, P, P, Z, P]),
    're2e_conv3x3_wino_wgrad_workspace_bytes': (Z, [I, I, I, I, I]),
    're2e_conv3x3_wino_wgrad': (I, [P, I, I, I, I, P, I, P, F, P, Z, P]),
    're2e_conv3x3_wino_rows': (I, [P, I, I, I, I, P, I, I, P, I, P, P, P, P, P, P, Z, P]),
    're2e_conv3x3_wino_wgrad_rows': (I, [P, I, I, I, I, P, I, P, F, P, P, Z, P]),
    're2e_fill_image_rows': (I, [P, I, I, L, P, I, I, F, P]),
    're2e_conv4x4_wino_workspace_bytes': (Z, [I, I, I, I, I, I]),
    're2e_conv4x4_wino': (I, [P, I, I, I, I, P, I, I, I, P, P, Z, P]),
    're2e_conv4x4_wino_wgrad_workspace_bytes': (Z, [I, I, I, I, I, I]),
    're2e_conv4x4_wino_wgrad': (I, [P, I, I, I, I, P, I, I, P, F, P, Z, P]),
    're2e_conv_wgrad_workspace_bytes': (Z, [I, I, I, I, I, I, I]),
    're2e_conv_wgrad': (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, I, I, P, F, P, Z, P]),
    're2e_conv_dgrad_s2': (I, [P, I, I, I, I, P, I, I, I, I, I, I, P, P, P]),
    're2e_conv_weight_gather': (I, [P, P, I, I, I, I, I, I, I, I, I, I, P]),
    're2e_transpose01': (I, [P, P, I, I, I, P]),
    're2e_act_fwd': (I, [P, P, L, I, P]),
    're2e_act_bwd': (I, [P, P, P, L, I, P]),
    're2e_act_bwd_colsum': (I, [P, P, P, I, I, I, P, F, P, Z, P]),
    're2e_colsum_workspace_bytes': (Z, [I, I]),
    're2e_colsum': (I, [P, I, I, L, P, F, P, Z, P]),
    're2e_mask_mul_bwd': (I, [P, P, P, P, L, P]),
    're2e_mask_mul_bwd_ld': (I, [P, P, P, P, L, I, I, P]),
    're2e_mul': (I, [P, P, P, L, P]),
    're2e_affine_cols': (I, [P, P, P, P, L, I, P]),
    're2e_dropout': (I, [P, P, L, F, c_ulonglong, c_uint, P]),
    're2e_axpby': (I, [F, P, F, P, L, P]),
    're2e_gather_rows': (I, [P, P, P, I, I, P]),
    're2e_scatter_rows': (I, [P, P, P, I, I, P]),
    're2e_mask_rows': (I, [P, P, P, I, I, I, P]),
    're2e_pack_pad': (I, [P, P, P, I, I, I, P, P]),
    're2e_kaldi_decode_pad': (I, [P, P, P, P, I, I, I, P, P, P, P]),
    're2e_fbank_fwd': (I, [P, L, I, I, P, P, P, I, P, P, P, P, P]),
    're2e_fbank_bwd': (I, [P, L, I, I, P, P, P, I, P, P, P, P, P, P]),
    're2e_logclamp_fwd': (I, [P, P, L, I, P, P]),
    're2e_logclamp_bwd': (I, [P, P, L, I, P, P, P]),
    're2e_cmvn_stats': (I, [P, P, I, I, I, P, P, P]),
    're2e_reduce_workspace_bytes': (Z, [L]),
    're2e_loss_fwd': (I, [P, P, F, L, I, P, P, Z, P]),
    're2e_loss_bwd': (I, [P, P, F, L, I, P, F, P, F, P]),
    're2e_sumsq': (I, [P, L, P, P, Z, P]),
    're2e_maxpool2_fwd': (I, [P, I, I, I, I, P, P, I, P]),
    're2e_maxpool2_bwd': (I, [P, P, I, I, I, I, P, P]),
    're2e_vgg_pack_fwd': (I, [P, P, I, I, I, I, P, I, I, P]),
    're2e_vgg_pack_bwd': (I, [P, P, I, I, I, I, P, I, I, P]),
    're2e_bn_workspace_bytes': (Z, [L, I]),
    're2e_bn_lrelu_fwd': (I, [P, L, I, P, P, P, P, F, F, I, F, P, P, P, P, Z, P]),
    're2e_bn_lrelu_bwd': (I, [P, P, L, I, P, P, P, P, F, P, P, P, F, P, Z, P]),
    're2e_bn_sync_partial': (I, [P, L, I, P, I, P, P, Z, P]),
    're2e_bn_sync_finalize': (I, [P, P, L, I, F, F, P, P, P, P, P]),
    're2e_bn_apply': (I, [P, L, I, P, P, P, P, F, P, P]),
    're2e_bn_sync_bwd_partial': (I, [P, P, L, I, P, P, P, P, F, P, P, Z, P]),
    're2e_bn_sync_bwd_apply': (I, [P, P, L, I, P, P, P, P, F, P, P, P]),
    're2e_lstm_workspace_bytes': (Z, [I, I]),
    're2e_lstm_abort_count': (I, []),
    're2e_debug_force_abort': (I, [I]),
    're2e_debug_occupy': (I, [I, I, I, P]),
    're2e_step_gate': (I, [P, I, P, P, P, P, P, P, P]),
    're2e_lstm_seq_fwd': (I, [P, P, P, P, P, P, P, I, I, I, P, Z, P]),
    're2e_lstm_seq_bwd': (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, P, Z, P]),
    're2e_lstm_cell_fwd': (I, [P, P, P, P, I, I, P]),
    're2e_lstm_cell_bwd': (I, [P, P, P, P, P, P, P, I, I, P]),
    're2e_dec_gates_cell_fwd': (I, [P, P, P, L, P, P, P, P, P, I, I, I, P]),
    're2e_gemm_skinny2': (I, [I, I, P, L, P, L, I, P, L, P, L, I, P, L, P]),
    're2e_dec_loop_workspace_bytes': (Z, [I, I, I, I, I, I, I, I]),
    're2e_dec_loop_bwd_workspace_bytes': (Z, [I, I, I, I, I, I, I, I]),
    're2e_dec_loop_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    're2e_dec_loop_dwconv': (I, [P, P, P, Z, P, I, I, I, I, I, I, I, I, I, I, P]),
    're2e_dec_loop_fwd': (I, [P, P, P, P, P, P, P, P, P, L, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    're2e_embedding_fwd': (I, [P, P, I, I, P, L, P]),
    're2e_embedding_bwd': (I, [P, L, P, I, I, I, P, F, P]),
    're2e_lsm_fwd': (I, [P, P, I, I, I, P, P, Z, P]),
    're2e_lsm_bwd': (I, [P, P, I, I, I, P, P, P]),
    're2e_log_softmax_rows': (I, [P, I, I, L, P, P]),
    're2e_argmax_rows': (I, [P, I, I, L, P, P]),
    're2e_ce_fwd': (I, [P, P, I, I, F, P, P, P, Z, P]),
    're2e_ce_bwd': (I, [P, P, P, P, I, I, F, P, P, P]),
    're2e_ctc_workspace_bytes': (Z, [I, I, I]),
    're2e_ctc_fwd': (I, [P, I, I, I, P, P, P, P, I, P, P, P, Z, P]),
    're2e_ctc_bwd': (I, [P, I, I, I, P, P, P, P, I, P, P, P, I, P, P]),
    're2e_ctc_prefix_score': (I, [P, I, I, P, I, P, P, P, P, I, F, F, I, I, P, P, P, P, P]),
    're2e_ctc_prefix_score_cands': (I, [P, I, I, P, I, P, P, P, P, P, I, F, F, I, I, P, P, P, P]),
    're2e_attloc_fwd': (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, L, P, P, P, P]),
    're2e_attloc_partial_floats': (Z, [I, I, I]),
    're2e_attloc_workspace_bytes': (Z, [I, I, I, I]),
    're2e_attloc_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, P, L, P, I, I, I, I, I, I, P, P, P, P, P, Z, P]),
    're2e_attloc_dpre': (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, Z, P]),
    're2e_attloc_denc': (I, [P, P, I, I, I, I, P, F, P]),
    're2e_clip_coef': (I, [P, F, P, P]),
    're2e_adadelta_step': (I, [P, P, P, P, L, F, F, F, P, P]),
    're2e_adam_step': (I, [P, P, P, P, L, F, F, F, F, I, P, P]),
}

def exp_env(name, default=None):
    """Host-side experiment / A-B switches (schedule variants, fusions turned off) are honoured only when RE2E_EXPERIMENTS=1 is set,
    like the library's own (csrc/common.h exp_env): a stray variable must not change how the shipped path runs.  The documented
    switches read directly are RE2E_LIB (another build of the same ABI), RE2E_NO_OVERLAP=1 (single-stream schedule, for per-kernel
    profiles; tests run both schedules) and RE2E_TIMELINE=1 (phase marks for tools/step_timeline.py)."""
    return os.environ.get(name, default) if os.environ.get('RE2E_EXPERIMENTS') == '1' else default


_lib = None


class Re2eError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Re2eError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.re2e_version.restype = ctypes.c_int
    got = lib.re2e_version()
    if got != ABI_VERSION:
        # the arguments are positional: a stale library (RE2E_LIB override, a tree that was not rebuilt) would take shifted
        # ints / floats / pointers without any error
        raise Re2eError('%s was built for ABI version %d, this binding is written for %d (include/re2e.h RE2E_ABI_VERSION): rebuild it '
                        'with `python -c "import __graft_entry__ as g; g.build()"`' % (LIB_PATH, got, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    try:
        if torch.cuda.is_available():
            lib.re2e_warmup()        # persistent recurrences: first sequence at full speed (best effort; a failure shows up at the first call)
    except Exception:
        pass
    return lib


def ptr(t):
    """Device pointer of a contiguous fp32/int32/uint8 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise Re2eError('expected a CUDA tensor, got device %s' % t.device)
    if not t.is_contiguous():
        raise Re2eError('expected a contiguous tensor')
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """The current torch stream of the current device as a raw hipStream_t.  ``torch.cuda.current_stream()`` builds a Stream object and
    re-derives the device index through ``is_available()`` every time (19 us a call, ~240 calls per training step: 4.6 ms of host time per
    step, tools/soak_step.py); the raw getter costs 1 us."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


FLOP_METER = None        # a dict while robust_e2e_gan_amd.flops.meter() is active: executed matrix-core FLOPs per entry point (bench.py)


def call(name, *args):
    """Invoke ``name`` on the current torch stream; raises Re2eError(re2e_last_error()) on failure."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise Re2eError('%s failed (%d): %s' % (name, rc, lib.re2e_last_error().decode()))
    if FLOP_METER is not None:
        from . import flops
        flops.count(name, args)


def call_supported(name, *args):
    """``call`` for entry points that may decline a shape: False on RE2E_EUNSUPPORTED (nothing was launched), raises on any other error."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc == EUNSUPPORTED:
        return False
    if rc != 0:
        raise Re2eError('%s failed (%d): %s' % (name, rc, lib.re2e_last_error().decode()))
    if FLOP_METER is not None:
        from . import flops
        flops.count(name, args)
    return True


def set_stream_role(torch_stream, filler=True):
    """Mark / unmark a torch stream as a FILLER stream (re2e_stream_role): bulk work that runs beside resident recurrences."""
    lib = load()
    rc = lib.re2e_stream_role(c_void_p(torch_stream.cuda_stream), STREAM_FILLER if filler else STREAM_DEFAULT)
    if rc != 0:
        raise Re2eError('re2e_stream_role failed (%d): %s' % (rc, lib.re2e_last_error().decode()))


def query(name, *args):
    return getattr(load(), name)(*args)


_STEP_STREAMS = {}


def step_streams(device=None, main_priority=-1):
    """(main, side, wgrad): ONE set of step streams per device and process, shared by every trainer (only one steps at a time).  main: the
    critical stream (high priority); side / wgrad: filler streams (re2e_stream_role).  Shared because a HIP stream is bound to one of the process's
    hardware queues when it is created (4 by default, 8 with GPU_MAX_HW_QUEUES=8), round-robin: the streams of the FIRST trainer of a process sit on
    three different queues, those of the fourth may not -- two of a step's streams on one queue serialise against each other (bench.py's
    other_configs leg, which builds four trainers in one process: configuration 3 at 24.0 instead of 21.7 ms per step, configuration 5 at 90.0
    instead of 85.1)."""
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    key = (idx, int(main_priority))
    got = _STEP_STREAMS.get(key)
    if got is None:
        with torch.cuda.device(idx):
            main = torch.cuda.Stream(priority=int(main_priority))
            side, wgrad = torch.cuda.Stream(), torch.cuda.Stream()
        set_stream_role(side, True)          # both run beside the resident recurrences of the main stream: 4-wave engine tiles there
        set_stream_role(wgrad, True)
        got = _STEP_STREAMS[key] = (main, side, wgrad)
    return got


_ws_cache = {}


def workspace(nbytes, device, tag='ws'):
    """Grow-only scratch buffer per (device, stream, tag); contents are undefined between calls.
    Keyed by the current stream so that work overlapped on a side stream never shares scratch."""
    nbytes = max(int(nbytes), 16)
    key = (str(device), stream() if torch.cuda.is_available() else 0, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4 + 1024, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


_hip = None


_masked_streams = []


def _destroy_masked_streams():
    """Registered only under rocprofv3 (its tool library is in LD_PRELOAD): streams created through
    hipExtStreamCreateWithCUMask that are still alive crash the profiler's finalisation, so they are drained and
    destroyed at exit.  NOT done otherwise: a tensor that outlives this handler (module-level objects, a failing test's
    traceback) would have the caching allocator record events on a destroyed stream."""
    try:
        _hip.hipStreamSynchronize.argtypes = [c_void_p]
        _hip.hipStreamDestroy.argtypes = [c_void_p]
        for h in _masked_streams:
            _hip.hipStreamSynchronize(c_void_p(h))
            _hip.hipStreamDestroy(c_void_p(h))
    except Exception:
        pass
    del _masked_streams[:]


def cu_masked_stream(enabled_cus, total_cus=256, device=None, first=0):
    """A HIP stream whose kernels may only run on bits ``first .. first + enabled_cus - 1`` of the CU mask
    (hipExtStreamCreateWithCUMask), wrapped as a torch ExternalStream.  Used for the filler streams of the
    joint step so that the short dependent launches of the recurrent chains always find idle CUs.
    Returns None if the runtime refuses."""
    global _hip
    try:
        if _hip is None:
            _hip = ctypes.CDLL('libamdhip64.so')
            _hip.hipExtStreamCreateWithCUMask.restype = c_int
            _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        words = (total_cus + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for i in range(first, min(first + enabled_cus, total_cus)):   # bit i = CU i // 8 of XCD i % 8 (tools/micro/cumask_probe.hip)
            mask[i // 32] |= (1 << (i % 32))
        st = c_void_p()
        rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
        if rc != 0 or not st.value:
            return None
        if not _masked_streams and 'rocprof' in os.environ.get('LD_PRELOAD', ''):
            atexit.register(_destroy_masked_streams)
        _masked_streams.append(st.value)
        return torch.cuda.ExternalStream(st.value, device=device)
    except Exception:
        return None
