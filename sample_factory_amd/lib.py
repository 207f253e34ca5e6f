"""ctypes binding of libsf_hip.so (C ABI: include/sf_hip.h).

PyTorch is plumbing only: it owns device memory (the caching allocator) and the stream; every call below hands raw
device pointers + the current HIP stream to the library.  There is no fallback: a missing library or a non-GPU tensor
raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_HIP_LIB: load another build of the same sources (kernel experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("SF_HIP_LIB") or os.path.join(_HERE, "libsf_hip.so")


class SfHipError(RuntimeError):
    pass


class sf_loss_cfg(C.Structure):
    _fields_ = [("clip_ratio", C.c_float), ("clip_value", C.c_float), ("value_loss_coeff", C.c_float),
                ("exploration_coeff", C.c_float), ("kl_coeff", C.c_float), ("exploration_kind", C.c_int32),
                ("action_kind", C.c_int32), ("dense_adv", C.c_int32), ("num_heads", C.c_int32),
                ("head_n", C.c_int32 * 8), ("old_values_T", C.c_int32)]


class sf_conv_desc(C.Structure):
    _fields_ = [("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32), ("KH", C.c_int32),
                ("KW", C.c_int32), ("stride", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
                ("in_u8", C.c_int32), ("relu", C.c_int32), ("traj_T", C.c_int32), ("sub_mean", C.c_float),
                ("inv_scale", C.c_float)]


# every symbol include/sf_hip.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    "sf_last_error", "sf_abi_version", "sf_valid_mask", "sf_gae_returns", "sf_moments", "sf_rms_update",
    "sf_rms_apply", "sf_vtrace", "sf_ppo_loss", "sf_loss_scalars", "sf_train_summaries", "sf_minibatch_indices", "sf_minibatch_expand", "sf_grad_sumsq",
    "sf_adam_step", "sf_adam_step_dlr", "sf_lr_kl_adaptive", "sf_clock_probe", "sf_lamb_step", "sf_rnn_cell_fwd", "sf_rnn_cell_bwd", "sf_rows_add_scale", "sf_mlp2_fwd", "sf_rnn_store_state", "sf_rnn_chunk_setup", "sf_lstm_seq_supported", "sf_lstm_seq_fwd", "sf_lstm_seq_bwd", "sf_gru_seq_fwd", "sf_gru_seq_bwd", "sf_seq_fwd_x_supported", "sf_lstm_seq_fwd_x", "sf_gru_seq_fwd_x", "sf_linear_fwd_dual_supported", "sf_linear_fwd_dual",
    "sf_obsnorm_moments", "sf_obsnorm_update", "sf_obsnorm_apply", "sf_sample_write_step",
    "sf_sample_write_step_tuple", "sf_sample_write_step_masked", "sf_traj_write_env_step", "sf_synth_obs",
    "sf_synth_step", "sf_synth_vec_step", "sf_h2d_rows", "sf_copy_rows", "sf_conv_fwd", "sf_conv_fwd_workspace", "sf_conv_wgrad_workspace", "sf_conv_wgrad",
    "sf_conv_dgrad", "sf_conv_norm_supported", "sf_conv_fwd_norm", "sf_conv_wgrad_norm", "sf_conv_relu_mask_supported", "sf_conv_fwd_relu_mask", "sf_conv_wgrad_relu_mask", "sf_conv_kernel_name", "sf_conv_fwd_t_supported", "sf_conv_fwd_t_workspace", "sf_conv_fwd_t", "sf_transpose",
    "sf_tanh_scale_fwd", "sf_tanh_scale_bwd",
    "sf_linear_fwd", "sf_linear_wgrad_workspace", "sf_linear_wgrad", "sf_linear_dgrad", "sf_relu_mask",
    "sf_dp_unique_id", "sf_dp_comm_create", "sf_dp_comm_destroy", "sf_dp_comm_info", "sf_allreduce_grads",
    "sf_dp_allreduce_f64", "sf_dp_broadcast", "sf_dp_oneshot_create", "sf_dp_oneshot_connect", "sf_dp_oneshot_allreduce_f32",
    "sf_dp_oneshot_allreduce_f64", "sf_dp_oneshot_status", "sf_dp_oneshot_destroy",
]

_lib: Optional[C.CDLL] = None

# ---- launch programs.  A rollout step of a small network is eight kernels of 5-50 us; issued through the wrappers below
# (argument checks, pointer conversions, the model's layer loop) the host needs ~20 us per launch and the GPU idles a third
# of the step (profiles/r06_p_c5_gaps.txt).  A LaunchProgram is the list of foreign calls one such step made, with their
# CONVERTED arguments: replaying it costs one ctypes call per launch.  What varies between replays (the sampler's Philox
# step, the policy version) is passed as a ctypes cell (c_uint32 / c_float) whose value the caller updates; everything
# else — pointers, strides, sizes, the stream — is constant under the key the caller files the program under.
_REC = threading.local()  # .rec: the _Recorder of the program this THREAD is recording (the learner thread is not affected)
_QUERIES = ("_supported", "_workspace", "_kernel_name", "sf_last_error", "sf_abi_version")  # no launch: never recorded
_HOST_STATE = ("sf_h2d_rows", "sf_dp_", "sf_allreduce_grads", "sf_clock_probe")  # read host memory / communicators: not replayable
LAUNCH_PROGRAMS = os.environ.get("SF_LAUNCH_PROGRAMS", "1") != "0"


class LaunchProgram:
    __slots__ = ("calls", "keep", "unsafe")

    def __init__(self):
        self.calls = []     # (foreign function, converted arguments, name, profiling key | None)
        self.keep = []      # every tensor whose address a call holds: its memory stays allocated as long as the program
        self.unsafe = None  # reason this recording must not be replayed (work outside the library was part of the step)

    def replay(self) -> None:
        if PROFILE is None:
            for fn, args, what, _ in self.calls:
                rc = fn(*args)
                if rc:
                    _check(rc, what)
        else:  # bench.py's per-launch HIP events see replayed launches like any other
            for fn, args, what, key in self.calls:
                with _timed(key):
                    _check(fn(*args), what)


class _Recorder:
    """stands where the CDLL stands while a program is recorded: every launch goes through AND is logged"""

    def __init__(self, real, prog: LaunchProgram):
        self._real, self.prog, self.key = real, prog, None

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name.endswith(_QUERIES):
            return fn
        if name.startswith(_HOST_STATE):
            self.prog.unsafe = self.prog.unsafe or name
            return fn

        def call(*args):
            self.prog.calls.append((fn, args, name, self.key))
            return fn(*args)
        return call


class record_launches:
    """with record_launches() as prog: ... — every library launch the block issues on this thread runs and is logged"""

    def __enter__(self) -> LaunchProgram:
        if getattr(_REC, "rec", None) is not None:
            raise SfHipError("record_launches: already recording on this thread")
        prog = LaunchProgram()
        real = load()
        _REC.rec = _Recorder(real, prog)
        return prog

    def __exit__(self, *a):
        _REC.rec = None
        return False


def recording_unsafe(reason: str) -> None:
    """called by code that is about to do work OUTSIDE the library (a torch op, a host copy) as part of a step: a program
    being recorded around it would silently drop that work on replay, so it is marked and never replayed"""
    rec = getattr(_REC, "rec", None)
    if rec is not None:
        rec.prog.unsafe = rec.prog.unsafe or reason


def _keep(t):
    rec = getattr(_REC, "rec", None)
    if rec is not None:
        rec.prog.keep.append(t)


def _vp(t: torch.Tensor) -> C.c_void_p:
    """address of a device view whose layout the caller has checked"""
    _keep(t)
    return C.c_void_p(t.data_ptr())


def _keys_wanted() -> bool:
    return PROFILE is not None or getattr(_REC, "rec", None) is not None

# Optional per-launch HIP-event timing of the network kernels (bench.py's roofline leg).  Events are recorded on
# torch's CURRENT stream, which is the stream every call below launches on.  PROFILE maps key -> [(start, end), ...].
PROFILE: Optional[dict] = None
PROFILE_ONLY: Optional[set] = None  # when set, only these keys are timed (bench: the dominant kernel only)
PROFILE_SYNC = False  # ranking pass: drain the device before every timed launch, so that an event pair brackets the
#                       kernel alone (with every launch instrumented the host falls behind and queueing delays would be
#                       charged to whatever small kernel happens to be launched next)


PROFILE_STRIDE = 1    # time every PROFILE_STRIDE-th launch of a key (bench.py's timed region: a HIP-event pair costs ~19 us of
#                       step time per timed launch; the average over a regular subsample is the same measurement)
PROFILE_SEEN: dict = {}  # key -> launches seen so far (timed or not)


class _timed:
    def __init__(self, key):
        self.rec = getattr(_REC, "rec", None)
        if self.rec is not None:
            self.rec.key = key  # the launches inside this block replay under the same key
        self.key = key if PROFILE is not None and key is not None and \
            (PROFILE_ONLY is None or key in PROFILE_ONLY) else None
        if self.key is not None and PROFILE_STRIDE > 1:
            seen = PROFILE_SEEN.get(self.key, 0)
            PROFILE_SEEN[self.key] = seen + 1
            if seen % PROFILE_STRIDE:
                self.key = None

    def __enter__(self):
        if self.key is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            if PROFILE_SYNC:
                torch.cuda.synchronize()
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.key is not None:
            self.e.record()
            PROFILE.setdefault(self.key, []).append((self.s, self.e))
        if self.rec is not None:
            self.rec.key = None
        return False


_OPS = {"fwd": 0, "wgrad": 1, "dgrad": 2, "fwd_t": 3}
_names: dict = {}


def _dkey(op, n, d):
    """profiling key of a network launch: (op, n, geometry..., kernel instantiation name as rocprofv3 prints it)"""
    if not _keys_wanted():
        return None
    k = (op, int(n), d.Cin, d.H, d.W, d.Cout, d.KH, d.stride, d.OH, d.OW, d.in_u8)
    name = _names.get(k)
    if name is None:
        name = _names[k] = conv_kernel_name(_OPS[op], n, d)
    return k[:-1] + (name,)


def conv_kernel_name(op: int, n: int, desc) -> str:
    buf = C.create_string_buffer(96)
    _check(load().sf_conv_kernel_name(int(op), i64(n), C.byref(desc), 1, buf, 96), "sf_conv_kernel_name")
    return buf.value.decode()


def load() -> C.CDLL:
    """Load the HIP library or fail loudly (no CPU path exists)."""
    global _lib
    rec = getattr(_REC, "rec", None)
    if rec is not None:
        return rec
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfHipError(
                f"{LIB_PATH} not found: build it with `python -m sample_factory_amd.build` (hipcc, gfx950). "
                "sample_factory_amd has no CPU/PyTorch fallback for the hot path.")
        lib = C.CDLL(LIB_PATH)
        lib.sf_last_error.restype = C.c_char_p
        lib.sf_conv_wgrad_workspace.restype = C.c_int64
        lib.sf_conv_fwd_workspace.restype = C.c_int64
        lib.sf_conv_fwd_t_workspace.restype = C.c_int64
        lib.sf_linear_wgrad_workspace.restype = C.c_int64
        _lib = lib
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise SfHipError(f"{what} failed ({rc}): {load().sf_last_error().decode()}")


_DT = {"f32": torch.float32, "f64": torch.float64, "i32": torch.int32, "u8": (torch.uint8, torch.bool)}


def ptr(t: Optional[torch.Tensor], kind: str, name: str = "tensor") -> C.c_void_p:
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    want = _DT[kind]
    ok = t.dtype in want if isinstance(want, tuple) else t.dtype == want
    if not ok:
        raise SfHipError(f"{name}: expected dtype {kind}, got {t.dtype}")
    if not t.is_cuda:
        raise SfHipError(f"{name}: tensor lives on {t.device}; the hot path only runs on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise SfHipError(f"{name}: tensor must be contiguous (shape {tuple(t.shape)}, strides {t.stride()})")
    return _vp(t)


def _raw(t: torch.Tensor, kind: str, name: str) -> C.c_void_p:
    """Device pointer of a possibly STRIDED view (the callee is told the stride explicitly)."""
    want = _DT[kind]
    if t.dtype != want:
        raise SfHipError(f"{name}: expected dtype {kind}, got {t.dtype}")
    if not t.is_cuda:
        raise SfHipError(f"{name}: tensor lives on {t.device}; the hot path only runs on the GPU (no CPU fallback)")
    return _vp(t)


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f(x) -> C.c_float:
    return x if isinstance(x, C.c_float) else C.c_float(float(x))  # (a c_float cell: a launch program's variable)


def i64(x) -> C.c_int64:
    return C.c_int64(int(x))


def u32(x) -> C.c_uint32:
    return x if isinstance(x, C.c_uint32) else C.c_uint32(int(x) & 0xFFFFFFFF)


# ------------------------------------------------------------------------------------------------ thin wrappers
def valid_mask(policy_id, policy_version, valids, actions, num_actions, log_prob_actions, my_policy_id, train_step,
               max_policy_lag, num_invalid, valids_flat=None) -> None:
    E, T = policy_id.shape
    if valids_flat is not None and valids_flat.numel() != E * T:
        raise SfHipError(f"valid_mask: valids_flat must hold E*T = {E * T} elements, got {valids_flat.numel()}")
    _check(load().sf_valid_mask(ptr(policy_id, "i32", "policy_id"), ptr(policy_version, "f32", "policy_version"),
                                ptr(valids, "u8", "valids"), ptr(valids_flat, "u8", "valids_flat"),
                                ptr(actions, "f32", "actions"), int(num_actions),
                                ptr(log_prob_actions, "f32", "log_prob_actions"), E, T, int(my_policy_id),
                                int(train_step), int(max_policy_lag), ptr(num_invalid, "i32", "num_invalid"),
                                stream()), "sf_valid_mask")


def gae_returns(rewards, dones, time_outs, values, valids, rms_stats, gamma, gae_lambda, value_bootstrap, advantages,
                returns) -> None:
    E, T = rewards.shape
    assert values.shape == (E, T + 1) and valids.shape == (E, T + 1)
    _check(load().sf_gae_returns(ptr(rewards, "f32", "rewards"), ptr(dones, "u8", "dones"),
                                 ptr(time_outs, "u8", "time_outs"), ptr(values, "f32", "values"),
                                 ptr(valids, "u8", "valids"), ptr(rms_stats, "f64", "rms_stats"), E, T, f(gamma),
                                 f(gae_lambda), int(bool(value_bootstrap)), ptr(advantages, "f32", "advantages"),
                                 ptr(returns, "f32", "returns"), stream()), "sf_gae_returns")


def moments(x, valids, index, n, out, offset=0, dense_x=False) -> None:
    """{sum, sumsq, count} of the minibatch rows j = index[i] | offset + i that are valid; x[j], or x[i] if dense_x"""
    _check(load().sf_moments(ptr(x, "f32", "x"), ptr(valids, "u8", "valids"), ptr(index, "i32", "index"), i64(offset),
                             i64(n), int(bool(dense_x)), ptr(out, "f64", "moments"), stream()), "sf_moments")


def rms_update(stats_in, mom, stats_out) -> None:
    _check(load().sf_rms_update(ptr(stats_in, "f64"), ptr(mom, "f64"), ptr(stats_out, "f64"), stream()),
           "sf_rms_update")


def rms_apply(x, stats, denormalize: bool) -> None:
    _check(load().sf_rms_apply(ptr(x, "f32", "x"), i64(x.numel()), ptr(stats, "f64"), int(bool(denormalize)),
                               stream()), "sf_rms_apply")


def _raw_any(t: torch.Tensor, u8: bool, name: str) -> C.c_void_p:
    want = torch.uint8 if u8 else torch.float32
    if t.dtype != want:
        raise SfHipError(f"{name}: expected {want}, got {t.dtype}")
    if not t.is_cuda:
        raise SfHipError(f"{name} lives on {t.device}; the hot path only runs on the GPU (no CPU fallback)")
    return _vp(t)


def obsnorm_moments(inp, u8, stride, index, offset, traj_T, n, D, sub_mean, inv_scale, s, ss) -> None:
    _check(load().sf_obsnorm_moments(_raw_any(inp, u8, "obs"), int(u8), i64(stride), ptr(index, "i32", "index"),
                                     i64(offset), int(traj_T), i64(n), int(D), f(sub_mean), f(inv_scale),
                                     ptr(s, "f64", "sum"), ptr(ss, "f64", "sumsq"), stream()), "sf_obsnorm_moments")


def obsnorm_update(mean, var, count_in, count_out, s, ss, n, D, mu_tab, rstd_tab) -> None:
    _check(load().sf_obsnorm_update(ptr(mean, "f64"), ptr(var, "f64"), ptr(count_in, "f64"), ptr(count_out, "f64"),
                                    ptr(s, "f64"), ptr(ss, "f64"), i64(n), int(D), ptr(mu_tab, "f32"),
                                    ptr(rstd_tab, "f32"), stream()), "sf_obsnorm_update")


def obsnorm_apply(inp, u8, stride, index, offset, traj_T, n, D, C_, HW, sub_mean, inv_scale, mu, rstd, out) -> None:
    _check(load().sf_obsnorm_apply(_raw_any(inp, u8, "obs"), int(u8), i64(stride), ptr(index, "i32", "index"),
                                   i64(offset), int(traj_T), i64(n), int(D), int(C_), int(HW), f(sub_mean),
                                   f(inv_scale), ptr(mu, "f32"), ptr(rstd, "f32"), ptr(out, "f32", "out"), stream()),
           "sf_obsnorm_apply")


def _rp(t, name="tensor"):
    """raw f32 device pointer of a possibly strided view (None -> NULL)"""
    return C.c_void_p(0) if t is None else _raw(t, "f32", name)


def rnn_cell_fwd(kind, gx, gh, h_prev, ld_h, c_prev, ld_c, keep, Cn, H, gates_out, h_out, c_out, h_next, c_next) -> None:
    _check(load().sf_rnn_cell_fwd(int(kind), ptr(gx, "f32", "gx"), ptr(gh, "f32", "gh"), _rp(h_prev, "h_prev"),
                                  i64(ld_h), _rp(c_prev, "c_prev"), i64(ld_c), ptr(keep, "f32", "keep"), int(Cn),
                                  int(H), ptr(gates_out, "f32"), ptr(h_out, "f32"), ptr(c_out, "f32"),
                                  ptr(h_next, "f32"), ptr(c_next, "f32"), stream()), "sf_rnn_cell_fwd")


def rnn_cell_bwd(kind, dh, dc_in, gates, h_prev, ld_h, c_prev, ld_c, c_out, Cn, H, dgx, dgh, dh_direct, dc_prev) -> None:
    _check(load().sf_rnn_cell_bwd(int(kind), ptr(dh, "f32", "dh"), ptr(dc_in, "f32"), ptr(gates, "f32", "gates"),
                                  _rp(h_prev, "h_prev"), i64(ld_h), _rp(c_prev, "c_prev"), i64(ld_c),
                                  ptr(c_out, "f32"), int(Cn), int(H), ptr(dgx, "f32", "dgx"), ptr(dgh, "f32"),
                                  ptr(dh_direct, "f32"), ptr(dc_prev, "f32"), stream()), "sf_rnn_cell_bwd")


def mlp2_supported(D, H1, H2) -> bool:
    return (H1 % 8 == 0 and H2 % 8 == 0 and (D * H1 + H1 * H2 + 64 * (D + H1)) * 4 <= 64 * 1024 and D <= 64 and
            D * H1 <= 8192 and H1 * H2 <= 8192)


def mlp2_fwd(x, x_stride, n, D, sub_mean, inv_scale, mu, rstd, w1, b1, w2, b2, act, out) -> None:
    _check(load().sf_mlp2_fwd(_raw(x, "f32", "x"), i64(x_stride), i64(n), int(D), f(sub_mean), f(inv_scale),
                              ptr(mu, "f32", "mu"), ptr(rstd, "f32", "rstd"), ptr(w1, "f32", "w1"), ptr(b1, "f32", "b1"),
                              int(w1.shape[1]), ptr(w2, "f32", "w2"), ptr(b2, "f32", "b2"), int(w2.shape[1]), int(act),
                              ptr(out, "f32", "out"), stream()), "sf_mlp2_fwd")


def rnn_store_state(h, c, dones_col, out) -> None:
    """out[b] = [h[b] | c[b]] * (1 - dones_col[b]); dones_col / out may be strided views of the slab"""
    B, H = h.shape
    if dones_col.dtype not in (torch.bool, torch.uint8) or not dones_col.is_cuda or dones_col.shape != (B,):
        raise SfHipError(f"rnn_store_state: dones must be a device bool/u8 [B] column, got {dones_col.dtype} {tuple(dones_col.shape)}")
    if out.shape != (B, H if c is None else 2 * H) or out.stride(1) != 1:
        raise SfHipError(f"rnn_store_state: out must be [B, {H if c is None else 2 * H}] with unit column stride")
    _check(load().sf_rnn_store_state(ptr(h, "f32", "h"), ptr(c, "f32", "c"), _vp(dones_col),
                                     i64(dones_col.stride(0)), _raw(out, "f32", "out"), i64(out.stride(0)), i64(B), int(H),
                                     stream()), "sf_rnn_store_state")


def rnn_chunk_setup(dones, valids, rnn_states, index, offset, Cn, R, keep_tm, h0, traj_T=0) -> None:
    """chunk-start states and done-or-invalid boundaries of a recurrent minibatch in one launch (see sf_hip.h);
    traj_T > 0: rnn_states is the slab [E, traj_T + 1, S], read in place"""
    S = rnn_states.shape[-1]
    if not rnn_states.is_contiguous() or keep_tm.shape != (R, Cn) or h0.shape != (Cn, S) or \
            (traj_T > 0 and (rnn_states.dim() != 3 or rnn_states.shape[1] != traj_T + 1)):
        raise SfHipError("rnn_chunk_setup: rnn_states must be contiguous [rows, S] (or the slab [E, T+1, S] with traj_T), "
                         "keep_tm [R, Cn], h0 [Cn, S]")
    _check(load().sf_rnn_chunk_setup(ptr(dones, "u8", "dones"), ptr(valids, "u8", "valids"),
                                     ptr(rnn_states, "f32", "rnn_states"), ptr(index, "i32"), i64(offset), int(Cn),
                                     int(R), int(S), int(traj_T), ptr(keep_tm, "f32", "keep_tm"), ptr(h0, "f32", "h0"),
                                     stream()),
           "sf_rnn_chunk_setup")


def lstm_seq_supported(Cn: int, H: int) -> bool:
    return bool(load().sf_lstm_seq_supported(int(Cn), int(H)))


def _seq_key(op, R, Cn, H, steps, G=4, x_cols=0):
    """profiling key of a fused LSTM pass in bench.py's layout: 2 * (steps*Cn) * 4H * H algorithmic FLOPs of the
    recurrent products (forward: R steps; backward: R-1, the first step has no state in front of it)"""
    if not _keys_wanted():
        return None
    kind, direction = op.split("_")
    name = None
    if direction == "bwd" and int(os.environ.get("SF_SEQ_BWD_REGW", "1")) and H in (256, 512):
        # csrc/sf_rnn_regw.h seq_plan_r: 32 hidden units per work-group, row groups of 32 / 64 rows
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        ng = max(1, min(cus // (H // 32), 16, (Cn + 31) // 32))
        rpg = ((Cn + ng - 1) // ng + 31) // 32 * 32
        if rpg <= 64:
            name = f"k_{kind}_seq_bwd_r<{int(H)}, {rpg // 16}>"
    if name is None:
        ng = min(8, (Cn + 15) // 16)
        rpg = ((Cn + ng - 1) // ng + 15) // 16 * 16
        nsub = {1: 1, 2: 2}.get((rpg + 63) // 64, 4)
        name = f"k_{kind}_seq_{direction}<{int(H)}, 16, {nsub}" + (f", {x_cols // 16}>" if direction == "fwd" else ">")
    # (K = H + x_cols: the fused input projection's FLOPs count as the pass's own)
    return (op, int(steps * Cn), int(H + x_cols), 1, 1, int(G * H), 1, 1, 1, 1, name)


def lstm_seq_fwd(gx, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H, env_major=False) -> None:
    with _timed(_seq_key("lstm_fwd", R, Cn, H, R)):
        _check(load().sf_lstm_seq_fwd(ptr(gx, "f32", "gx"), ptr(whh, "f32", "whh"), ptr(bhh, "f32", "bhh"),
                                      ptr(keep, "f32", "keep"), ptr(gates, "f32", "gates"), ptr(hprev, "f32", "hprev"),
                                      ptr(hout, "f32", "hout"), ptr(cprev, "f32", "cprev"), ptr(cout, "f32", "cout"),
                                      ptr(sync, "i32", "sync"), int(R), int(Cn), int(H), int(bool(env_major)), stream()),
               "sf_lstm_seq_fwd")


def lstm_seq_bwd(dout, gates, cprev, cout, keep, whh, dgx, sync, R, Cn, H, env_major=False) -> None:
    with _timed(_seq_key("lstm_bwd", R, Cn, H, R - 1)):
        _check(load().sf_lstm_seq_bwd(ptr(dout, "f32", "dout"), ptr(gates, "f32", "gates"), ptr(cprev, "f32", "cprev"),
                                      ptr(cout, "f32", "cout"), ptr(keep, "f32", "keep"), ptr(whh, "f32", "whh"),
                                      ptr(dgx, "f32", "dgx"),
                                      ptr(sync, "i32", "sync"), int(R), int(Cn), int(H), int(bool(env_major)), stream()),
               "sf_lstm_seq_bwd")


def gru_seq_fwd(gx, whh, bhh, keep, gates, hprev, hout, sync, R, Cn, H, env_major=False) -> None:
    with _timed(_seq_key("gru_fwd", R, Cn, H, R, 3)):
        _check(load().sf_gru_seq_fwd(ptr(gx, "f32", "gx"), ptr(whh, "f32", "whh"), ptr(bhh, "f32", "bhh"),
                                     ptr(keep, "f32", "keep"), ptr(gates, "f32", "gates"), ptr(hprev, "f32", "hprev"),
                                     ptr(hout, "f32", "hout"), ptr(sync, "i32", "sync"), int(R), int(Cn), int(H),
                                     int(bool(env_major)), stream()), "sf_gru_seq_fwd")


def linear_fwd_dual_supported(n: int, N: int, K1: int, K2: int) -> bool:
    return bool(load().sf_linear_fwd_dual_supported(i64(n), int(N), int(K1), int(K2)))


def linear_fwd_dual(a1, lda1, w1t, bias1, a2, lda2, w2t, bias2, out, n, gru_H=0) -> None:
    """out [n, N] = a1 w1t^T + a2 w2t^T + bias1 + bias2 (a_i: row-strided views, w_it: [N, K_i]); gru_H > 0: the GRU layout
    (w_it [3H, K_i], out [n, 4H], see sf_hip.h)"""
    N, K1, K2 = int(out.shape[1]), int(w1t.shape[1]), int(w2t.shape[1])
    # (GRU layout: 3H columns' worth of products spread over 4H output columns)
    key = None if not _keys_wanted() else ("fwd_dual", int(n), K1 + K2, 1, 1, 3 * int(gru_H) if gru_H else N, 1, 1, 1, 1, "k_fwd_glds2<128, 64, 2, 2>")
    with _timed(key):
        _check(load().sf_linear_fwd_dual(_raw(a1, "f32", "a1"), i64(lda1), ptr(w1t, "f32", "w1t"), ptr(bias1, "f32", "bias1"), K1,
                                         _raw(a2, "f32", "a2"), i64(lda2), ptr(w2t, "f32", "w2t"), ptr(bias2, "f32", "bias2"), K2,
                                         ptr(out, "f32", "out"), i64(n), N, int(gru_H), stream()), "sf_linear_fwd_dual")


def seq_fwd_x_supported(Cn: int, H: int, Kx: int) -> bool:
    return bool(load().sf_seq_fwd_x_supported(int(Cn), int(H), int(Kx)))


def lstm_seq_fwd_x(x, wih_t, bih, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H, env_major=False) -> None:
    """sf_lstm_seq_fwd with the input projection fused in: x [R*Cn, Kx] time-major, wih_t [4H, Kx], bih [4H]"""
    Kx = int(wih_t.shape[1])
    with _timed(_seq_key("lstm_fwd", R, Cn, H, R, x_cols=Kx)):
        _check(load().sf_lstm_seq_fwd_x(ptr(x, "f32", "x"), ptr(wih_t, "f32", "wih_t"), ptr(bih, "f32", "bih"), Kx,
                                        ptr(whh, "f32", "whh"), ptr(bhh, "f32", "bhh"), ptr(keep, "f32", "keep"),
                                        ptr(gates, "f32", "gates"), ptr(hprev, "f32", "hprev"), ptr(hout, "f32", "hout"),
                                        ptr(cprev, "f32", "cprev"), ptr(cout, "f32", "cout"), ptr(sync, "i32", "sync"),
                                        int(R), int(Cn), int(H), int(bool(env_major)), stream()), "sf_lstm_seq_fwd_x")


def gru_seq_fwd_x(x, wih_t, bih, whh, bhh, keep, gates, hprev, hout, sync, R, Cn, H, env_major=False) -> None:
    Kx = int(wih_t.shape[1])
    with _timed(_seq_key("gru_fwd", R, Cn, H, R, 3, x_cols=Kx)):
        _check(load().sf_gru_seq_fwd_x(ptr(x, "f32", "x"), ptr(wih_t, "f32", "wih_t"), ptr(bih, "f32", "bih"), Kx,
                                       ptr(whh, "f32", "whh"), ptr(bhh, "f32", "bhh"), ptr(keep, "f32", "keep"),
                                       ptr(gates, "f32", "gates"), ptr(hprev, "f32", "hprev"), ptr(hout, "f32", "hout"),
                                       ptr(sync, "i32", "sync"), int(R), int(Cn), int(H), int(bool(env_major)), stream()),
               "sf_gru_seq_fwd_x")


def gru_seq_bwd(dout, gates, hprev, keep, whh, dgx, dgh, sync, R, Cn, H, env_major=False) -> None:
    with _timed(_seq_key("gru_bwd", R, Cn, H, R - 1, 3)):
        _check(load().sf_gru_seq_bwd(ptr(dout, "f32", "dout"), ptr(gates, "f32", "gates"), ptr(hprev, "f32", "hprev"),
                                     ptr(keep, "f32", "keep"), ptr(whh, "f32", "whh"), ptr(dgx, "f32", "dgx"),
                                     ptr(dgh, "f32", "dgh"), ptr(sync, "i32", "sync"), int(R), int(Cn), int(H),
                                     int(bool(env_major)), stream()), "sf_gru_seq_bwd")


def rows_add_scale(a, b, keep, Cn, H, y) -> None:
    _check(load().sf_rows_add_scale(ptr(a, "f32", "a"), ptr(b, "f32"), ptr(keep, "f32"), i64(Cn), int(H),
                                    ptr(y, "f32", "y"), stream()), "sf_rows_add_scale")


def vtrace(params, ld_params, values, ld_values, actions, old_logp, rewards, dones, index, offset, n, A, action_kind,
           recurrence, gamma, rho_hat, c_hat, vs, adv, head_sizes=None) -> None:
    hn = (C.c_int32 * 8)(*[int(x) for x in head_sizes]) if head_sizes and len(head_sizes) > 1 else None
    _check(load().sf_vtrace(_raw(params, "f32", "params"), int(ld_params), _raw(values, "f32", "values"),
                            int(ld_values), ptr(actions, "f32", "actions"),
                            ptr(old_logp, "f32", "old_logp"), ptr(rewards, "f32", "rewards"), ptr(dones, "u8", "dones"),
                            ptr(index, "i32", "index"), i64(offset), i64(n), int(A), int(action_kind),
                            int(recurrence), f(gamma), f(rho_hat), f(c_hat), ptr(vs, "f32", "vs"),
                            ptr(adv, "f32", "adv"), hn, len(head_sizes) if hn is not None else 0, stream()),
           "sf_vtrace")


def ppo_loss(params, ld_params, values, ld_values, actions, old_logp, old_params, old_values, adv, targets, valids,
             index, offset, n, A, cfg: sf_loss_cfg, mom, sums, g_params, g_values, ratio_out=None) -> None:
    """params/values (and g_params/g_values) may be strided column views of one [n, ld] matrix."""
    _check(load().sf_ppo_loss(_raw(params, "f32", "params"), int(ld_params), _raw(values, "f32", "values"),
                              int(ld_values), ptr(actions, "f32", "actions"), ptr(old_logp, "f32", "old_logp"),
                              ptr(old_params, "f32", "old_params"), ptr(old_values, "f32", "old_values"),
                              ptr(adv, "f32", "adv"), ptr(targets, "f32", "targets"), ptr(valids, "u8", "valids"),
                              ptr(index, "i32", "index"), i64(offset), i64(n), int(A), C.byref(cfg),
                              ptr(mom, "f64", "moments"), ptr(sums, "f64", "sums"), _raw(g_params, "f32", "g_params"),
                              _raw(g_values, "f32", "g_values"), ptr(ratio_out, "f32", "ratio_out"), stream()),
           "sf_ppo_loss")


def loss_scalars(sums, mom, cfg: sf_loss_cfg, out) -> None:
    _check(load().sf_loss_scalars(ptr(sums, "f64"), ptr(mom, "f64"), C.byref(cfg), ptr(out, "f32"), stream()),
           "sf_loss_scalars")


def train_summaries(valids, ratio, values, ld_values, old_values, old_values_T, actions, num_actions, adv, dense_adv,
                    policy_id, policy_version, action_logits, A, index, offset, n, my_policy_id, train_step, clip_ratio,
                    exp_avg_sq, out) -> None:
    """learner.py:843-923 for one minibatch in one pass; out: device double[24] (layout: include/sf_hip.h)"""
    _check(load().sf_train_summaries(ptr(valids, "u8", "valids"), ptr(ratio, "f32", "ratio"), _raw(values, "f32", "values"),
                                     int(ld_values), ptr(old_values, "f32", "old_values"), int(old_values_T),
                                     ptr(actions, "f32", "actions"), int(num_actions), ptr(adv, "f32", "adv"),
                                     int(bool(dense_adv)), ptr(policy_id, "i32", "policy_id"),
                                     ptr(policy_version, "f32", "policy_version"), ptr(action_logits, "f32", "action_logits"),
                                     int(A), ptr(index, "i32", "index"), i64(offset), i64(n), int(my_policy_id),
                                     int(train_step), f(clip_ratio), ptr(exp_avg_sq, "f32", "exp_avg_sq"),
                                     i64(exp_avg_sq.numel() if exp_avg_sq is not None else 0), ptr(out, "f64", "out"),
                                     stream()), "sf_train_summaries")


def minibatch_indices(out, experience_size, recurrence, shuffle, seed, epoch) -> None:
    _check(load().sf_minibatch_indices(ptr(out, "i32", "indices"), i64(experience_size), int(recurrence),
                                       int(bool(shuffle)), u32(seed), u32(epoch), stream()), "sf_minibatch_indices")


def minibatch_expand(chunk_starts, out, experience_size, recurrence) -> None:
    _check(load().sf_minibatch_expand(ptr(chunk_starts, "i32", "chunk_starts"), ptr(out, "i32", "out"),
                                      i64(experience_size), int(recurrence), stream()), "sf_minibatch_expand")


def grad_sumsq(g, sumsq) -> None:
    _check(load().sf_grad_sumsq(ptr(g, "f32", "grad"), i64(g.numel()), ptr(sumsq, "f64", "sumsq"), stream()),
           "sf_grad_sumsq")


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, max_grad_norm, sumsq, grad_scale=1.0, skip_flag=None) -> None:
    """skip_flag: device int32 [1] view (the sticky abort word of the fused recurrent passes) or None"""
    _check(load().sf_adam_step(ptr(p, "f32", "params"), ptr(g, "f32", "grads"), ptr(m, "f32", "exp_avg"),
                               ptr(v, "f32", "exp_avg_sq"), i64(p.numel()), int(step), f(lr), f(beta1), f(beta2),
                               f(eps), f(max_grad_norm), ptr(sumsq, "f64", "sumsq"), f(grad_scale),
                               ptr(skip_flag, "i32", "skip_flag"), stream()),
           "sf_adam_step")


def adam_step_dlr(p, g, m, v, step, lr_dev, lr_scale, beta1, beta2, eps, max_grad_norm, sumsq, grad_scale=1.0,
                  skip_flag=None) -> None:
    """sf_adam_step with lr = lr_dev[0] * lr_scale read on the device (KL-adaptive per-minibatch schedule without a sync)"""
    _check(load().sf_adam_step_dlr(ptr(p, "f32", "params"), ptr(g, "f32", "grads"), ptr(m, "f32", "exp_avg"),
                                   ptr(v, "f32", "exp_avg_sq"), i64(p.numel()), int(step), ptr(lr_dev, "f32", "lr_dev"),
                                   f(lr_scale), f(beta1), f(beta2), f(eps), f(max_grad_norm), ptr(sumsq, "f64", "sumsq"),
                                   f(grad_scale), ptr(skip_flag, "i32", "skip_flag"), stream()), "sf_adam_step_dlr")


def lr_kl_adaptive(kl, lr_dev, threshold, lr_min, lr_max, lr_out=None) -> None:
    """kl: device f32 [1] view (mean KL of the SGD step); lr_dev: device f32 [1], updated in place (learner.py:46-85)"""
    _check(load().sf_lr_kl_adaptive(ptr(kl, "f32", "kl"), ptr(lr_dev, "f32", "lr_dev"), f(threshold), f(lr_min), f(lr_max),
                                    ptr(lr_out, "f32", "lr_out"), stream()), "sf_lr_kl_adaptive")


def clock_probe(out: torch.Tensor, spin_cycles: int = 200000, on_stream=None) -> None:
    """out: device int64 [2] -> (shader cycles, 100 MHz wall-clock ticks) of a one-wave spin (bench.py roofline.clock_ghz)"""
    if not out.is_cuda or out.dtype != torch.int64 or out.numel() < 2:
        raise SfHipError("clock_probe: out must be a device int64 tensor of 2 elements")
    st = C.c_void_p(on_stream.cuda_stream) if on_stream is not None else stream()
    _check(load().sf_clock_probe(C.c_void_p(out.data_ptr()), int(spin_cycles), st), "sf_clock_probe")


def lamb_step(p, g, m, v, scratch, seg_id, seg_sums, num_segments, step, lr, beta1, beta2, eps, weight_decay, min_trust,
              max_grad_norm, sumsq, grad_scale=1.0, skip_flag=None) -> None:
    _check(load().sf_lamb_step(ptr(p, "f32", "p"), ptr(g, "f32", "g"), ptr(m, "f32", "m"), ptr(v, "f32", "v"),
                               ptr(scratch, "f32", "scratch"), ptr(seg_id, "u8", "seg_id"), ptr(seg_sums, "f64", "seg_sums"),
                               i64(p.numel()), int(num_segments), int(step), f(lr), f(beta1), f(beta2), f(eps),
                               f(weight_decay), f(min_trust), f(max_grad_norm), ptr(sumsq, "f64", "sumsq"),
                               f(grad_scale), ptr(skip_flag, "i32", "skip_flag"), stream()), "sf_lamb_step")


def sample_write_step(logits, ld_logits, values, ld_values, B, A, T, t, seed, step, row0, policy_version, deterministic,
                      traj_actions, traj_logits, traj_logp, traj_values, traj_policy_version, env_actions,
                      action_kind=0) -> None:
    _check(load().sf_sample_write_step(_raw(logits, "f32", "logits"), int(ld_logits), _raw(values, "f32", "values"),
                                       int(ld_values), int(B), int(A), int(T),
                                       int(t), u32(seed), u32(step), u32(row0), f(policy_version),
                                       int(bool(deterministic)), int(action_kind), ptr(traj_actions, "f32"),
                                       ptr(traj_logits, "f32"),
                                       ptr(traj_logp, "f32"), ptr(traj_values, "f32"),
                                       ptr(traj_policy_version, "f32"), ptr(env_actions, "i32"), stream()),
           "sf_sample_write_step")


def sample_write_step_masked(logits, ld_logits, values, ld_values, mask, ld_mask, B, A, T, t, seed, step, row0,
                             policy_version, deterministic, traj_actions, traj_logits, traj_logp, traj_values,
                             traj_policy_version, env_actions) -> None:
    """mask: u8/bool [B, A] view with row stride ld_mask (e.g. the slab's obs["action_mask"][:, t])"""
    if mask.dtype not in (torch.uint8, torch.bool) or not mask.is_cuda:
        raise SfHipError(f"action_mask: expected a u8/bool CUDA tensor, got {mask.dtype} on {mask.device}")
    _check(load().sf_sample_write_step_masked(_raw(logits, "f32", "logits"), int(ld_logits),
                                              _raw(values, "f32", "values"), int(ld_values),
                                              _vp(mask), i64(ld_mask), int(B), int(A), int(T), int(t),
                                              u32(seed), u32(step), u32(row0), f(policy_version),
                                              int(bool(deterministic)), ptr(traj_actions, "f32"),
                                              ptr(traj_logits, "f32"), ptr(traj_logp, "f32"), ptr(traj_values, "f32"),
                                              ptr(traj_policy_version, "f32"), ptr(env_actions, "i32"), stream()),
           "sf_sample_write_step_masked")


def sample_write_step_tuple(logits, ld_logits, values, ld_values, B, head_n, T, t, seed, step, row0, policy_version,
                            deterministic, traj_actions, traj_logits, traj_logp, traj_values, traj_policy_version,
                            env_actions) -> None:
    """Tuple of Discrete heads; env_actions int32 [B, len(head_n)]"""
    hn = (C.c_int32 * 8)(*[int(x) for x in head_n])
    _check(load().sf_sample_write_step_tuple(_raw(logits, "f32", "logits"), int(ld_logits),
                                             _raw(values, "f32", "values"), int(ld_values), int(B), len(head_n), hn,
                                             int(T), int(t), u32(seed), u32(step), u32(row0), f(policy_version),
                                             int(bool(deterministic)), ptr(traj_actions, "f32"), ptr(traj_logits, "f32"),
                                             ptr(traj_logp, "f32"), ptr(traj_values, "f32"),
                                             ptr(traj_policy_version, "f32"), ptr(env_actions, "i32"), stream()),
           "sf_sample_write_step_tuple")


def traj_write_env_step(rewards, terminated, truncated, T, t, reward_scale, reward_clip, policy_id, traj_rewards,
                        traj_dones, traj_time_outs, traj_policy_id, ep_return, ep_len, ep_stats) -> None:
    B = rewards.numel()
    _check(load().sf_traj_write_env_step(ptr(rewards, "f32", "rewards"), ptr(terminated, "u8", "terminated"),
                                         ptr(truncated, "u8", "truncated"), B, int(T), int(t), f(reward_scale),
                                         f(reward_clip), int(policy_id), ptr(traj_rewards, "f32"),
                                         ptr(traj_dones, "u8"), ptr(traj_time_outs, "u8"), ptr(traj_policy_id, "i32"),
                                         ptr(ep_return, "f32"), ptr(ep_len, "i32"), ptr(ep_stats, "f64"), stream()),
           "sf_traj_write_env_step")


def synth_vec_step(state, actions, obs_out, env0, seed, step, reset, rewards, terminated) -> None:
    """state [B, D] contiguous; actions [B, A] view (row stride), obs_out [B, D] view (row stride); see sf_hip.h"""
    B, D = state.shape
    A = actions.shape[1] if actions is not None else 0
    _check(load().sf_synth_vec_step(
        ptr(state, "f32", "state"), _raw(actions, "f32", "actions") if actions is not None else C.c_void_p(0),
        i64(actions.stride(0) if actions is not None else 0), _raw(obs_out, "f32", "obs_out"), i64(obs_out.stride(0)),
        int(B), int(D), int(A), int(env0), u32(seed), u32(step), int(bool(reset)),
        ptr(rewards, "f32", "rewards"), ptr(terminated, "u8", "terminated"), stream()), "sf_synth_vec_step")


def h2d_rows(dst: torch.Tensor, src_pinned: torch.Tensor) -> None:
    """dst: device view [rows, ...] whose rows are contiguous (any row pitch, e.g. slab[:, t]); src_pinned: contiguous
    pinned host tensor of the same shape/dtype.  One pitched DMA on the current stream."""
    if not dst.is_cuda or src_pinned.is_cuda or not src_pinned.is_pinned() or not src_pinned.is_contiguous():
        raise SfHipError("h2d_rows: dst must be a device view, src a contiguous pinned host tensor")
    if dst.shape != src_pinned.shape or dst.dtype != src_pinned.dtype or (dst.dim() > 1 and not dst[0].is_contiguous()):
        raise SfHipError(f"h2d_rows: shape/dtype/row layout mismatch {tuple(dst.shape)} {dst.dtype} vs "
                         f"{tuple(src_pinned.shape)} {src_pinned.dtype}")
    if dst.is_contiguous():  # one flat row
        rows, row_bytes = 1, dst.numel() * dst.element_size()
        pitch = row_bytes
    else:
        rows = dst.shape[0]
        row_bytes = (dst[0].numel() if dst.dim() > 1 else 1) * dst.element_size()
        pitch = dst.stride(0) * dst.element_size()
    _check(load().sf_h2d_rows(C.c_void_p(dst.data_ptr()), i64(pitch), C.c_void_p(src_pinned.data_ptr()), i64(row_bytes),
                              i64(row_bytes), i64(rows), stream()), "sf_h2d_rows")


def copy_rows(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst <- src for two device views of equal shape / dtype whose rows (dim 0) are contiguous blocks, any row pitch on
    either side (slab[:, t] columns, a column of a matrix): ONE launch of the library's own row-copy kernel"""
    if not (dst.is_cuda and src.is_cuda) or dst.shape != src.shape or dst.dtype != src.dtype:
        raise SfHipError(f"copy_rows: device views of equal shape/dtype required, got {tuple(dst.shape)} {dst.dtype} on "
                         f"{dst.device} <- {tuple(src.shape)} {src.dtype} on {src.device}")
    if dst.numel() == 0:
        return
    es = dst.element_size()

    def rows_of(t):
        if t.is_contiguous():
            return 1, t.numel() * es, t.numel() * es
        if t.dim() >= 1 and (t.dim() == 1 or t[0].is_contiguous()):
            return t.shape[0], (t[0].numel() if t.dim() > 1 else 1) * es, t.stride(0) * es
        raise SfHipError(f"copy_rows: rows must be contiguous blocks (shape {tuple(t.shape)}, strides {t.stride()})")
    rd, bd, pd = rows_of(dst)
    rs, bs, ps = rows_of(src)
    if (rd, bd) != (rs, bs):  # one side is fully contiguous: express it in the other side's row structure
        if rd == 1 and rs > 1:
            rd, bd, pd = rs, bs, bs
        elif rs == 1 and rd > 1:
            rs, bs, ps = rd, bd, bd
    _check(load().sf_copy_rows(_vp(dst), i64(pd), _vp(src), i64(ps), i64(bd),
                               i64(rd), stream()), "sf_copy_rows")


def synth_obs(obs_slot_ptr: int, env_stride, B, env0, obs_bytes, seed, step) -> None:
    _check(load().sf_synth_obs(C.c_void_p(obs_slot_ptr), i64(env_stride), int(B), int(env0), i64(obs_bytes),
                               u32(seed), u32(step), stream()), "sf_synth_obs")


def synth_step(actions, env0, num_actions, seed, step, rewards, terminated) -> None:
    _check(load().sf_synth_step(ptr(actions, "i32", "actions"), actions.numel(), int(env0), int(num_actions),
                                u32(seed), u32(step), ptr(rewards, "f32"), ptr(terminated, "u8"), stream()),
           "sf_synth_step")


# ---- network kernels (csrc/sf_nn.hip)
def _raw_in(t: torch.Tensor, desc: sf_conv_desc) -> C.c_void_p:
    want = torch.uint8 if desc.in_u8 else torch.float32
    if t.dtype != want:
        raise SfHipError(f"conv input: expected {want}, got {t.dtype}")
    if not t.is_cuda:
        raise SfHipError(f"conv input lives on {t.device}; the hot path only runs on the GPU (no CPU fallback)")
    return _vp(t)


def conv_fwd_workspace(n, desc: sf_conv_desc) -> int:
    return int(load().sf_conv_fwd_workspace(i64(n), C.byref(desc)))


def conv_fwd_raw(inp, in_sample_stride, index, offset, w, bias, out, n, desc: sf_conv_desc, workspace=None) -> None:
    """`inp` may be a strided view (e.g. slab[:, t]); its data_ptr is sample 0, samples are in_sample_stride apart."""
    with _timed(_dkey("fwd", n, desc)):
        _check(load().sf_conv_fwd(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"),
                                i64(offset), ptr(w, "f32", "w"), ptr(bias, "f32", "bias"), ptr(out, "f32", "out"),
                                i64(n), C.byref(desc), ptr(workspace, "u8", "workspace"),
                                i64(workspace.numel() if workspace is not None else 0), stream()), "sf_conv_fwd")


def conv_wgrad_workspace(n, desc: sf_conv_desc) -> int:
    return int(load().sf_conv_wgrad_workspace(i64(n), C.byref(desc)))


def conv_wgrad_raw(inp, in_sample_stride, index, offset, dout, dw, db, n, desc: sf_conv_desc, workspace) -> None:
    with _timed(_dkey("wgrad", n, desc)):
        _check(load().sf_conv_wgrad(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"),
                                  i64(offset), ptr(dout, "f32", "dout"), ptr(dw, "f32", "dw"), ptr(db, "f32", "db"),
                                  i64(n), C.byref(desc), ptr(workspace, "u8", "workspace"), stream()), "sf_conv_wgrad")


def conv_relu_mask_supported(n, desc: sf_conv_desc) -> bool:
    return bool(load().sf_conv_relu_mask_supported(i64(n), C.byref(desc)))


def conv_fwd_relu_mask(inp, in_sample_stride, index, offset, w, bias, out, relu_mask, n, desc: sf_conv_desc) -> None:
    """sf_conv_fwd + the ReLU sign bits of the output: relu_mask int32 [n * OH * OW] (one word per output pixel)"""
    with _timed(_dkey("fwd", n, desc)):
        _check(load().sf_conv_fwd_relu_mask(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"),
                                            i64(offset), ptr(w, "f32", "w"), ptr(bias, "f32", "bias"),
                                            ptr(out, "f32", "out"), ptr(relu_mask, "i32", "relu_mask"), i64(n),
                                            C.byref(desc), stream()), "sf_conv_fwd_relu_mask")


def conv_wgrad_relu_mask(inp, in_sample_stride, index, offset, dout, relu_mask, dw, db, n, desc: sf_conv_desc,
                         workspace) -> None:
    """sf_conv_wgrad on an UNMASKED output gradient: the recorded sign bits are applied inside the kernel"""
    with _timed(_dkey("wgrad", n, desc)):
        _check(load().sf_conv_wgrad_relu_mask(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"),
                                              i64(offset), ptr(dout, "f32", "dout"), ptr(relu_mask, "i32", "relu_mask"),
                                              ptr(dw, "f32", "dw"), ptr(db, "f32", "db"), i64(n), C.byref(desc),
                                              ptr(workspace, "u8", "workspace"), stream()), "sf_conv_wgrad_relu_mask")


def conv_norm_supported(n, desc: sf_conv_desc) -> bool:
    return bool(load().sf_conv_norm_supported(i64(n), C.byref(desc)))


def _nkey(op, n, desc, name):
    k = _dkey(op, n, desc)
    return None if k is None else k[:-1] + (name,)


def conv_fwd_norm(inp, in_sample_stride, index, offset, mu, rstd, w, bias, out, n, desc: sf_conv_desc) -> None:
    """conv1 on raw u8 frames with the observation normaliser's tables applied in the loader (sf_conv_fwd_norm)"""
    with _timed(_nkey("fwd", n, desc, "k_conv_u8_img_norm<2, 4, 5, 16>")):
        _check(load().sf_conv_fwd_norm(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"), i64(offset),
                                       ptr(mu, "f32", "mu"), ptr(rstd, "f32", "rstd"), ptr(w, "f32", "w"),
                                       ptr(bias, "f32", "bias"), ptr(out, "f32", "out"), i64(n), C.byref(desc),
                                       stream()), "sf_conv_fwd_norm")


def conv_wgrad_norm(inp, in_sample_stride, index, offset, mu, rstd, dout, dw, db, n, desc: sf_conv_desc, workspace) -> None:
    with _timed(_nkey("wgrad", n, desc, "k_conv1_wgrad_img_norm<2, 4>")):
        _check(load().sf_conv_wgrad_norm(_raw_in(inp, desc), i64(in_sample_stride), ptr(index, "i32", "index"),
                                         i64(offset), ptr(mu, "f32", "mu"), ptr(rstd, "f32", "rstd"),
                                         ptr(dout, "f32", "dout"), ptr(dw, "f32", "dw"), ptr(db, "f32", "db"), i64(n),
                                         C.byref(desc), ptr(workspace, "u8", "workspace"), stream()), "sf_conv_wgrad_norm")


def conv_dgrad(dout, w, in_act, din, n, desc: sf_conv_desc) -> None:
    with _timed(_dkey("dgrad", n, desc)):
        _check(load().sf_conv_dgrad(ptr(dout, "f32", "dout"), ptr(w, "f32", "w"), ptr(in_act, "f32", "in_act"),
                                  ptr(din, "f32", "din"), i64(n), C.byref(desc), stream()), "sf_conv_dgrad")


def conv_fwd_t_supported(n, desc: sf_conv_desc) -> bool:
    return bool(load().sf_conv_fwd_t_supported(i64(n), C.byref(desc)))


def conv_fwd_t_workspace(n, desc: sf_conv_desc) -> int:
    return int(load().sf_conv_fwd_t_workspace(i64(n), C.byref(desc)))


def conv_fwd_t(inp, in_sample_stride, wt, bias, out, n, desc: sf_conv_desc, workspace=None) -> None:
    """glds forward: wt is the [Cout, K] transpose of the canonical weights; workspace: conv_fwd_t_workspace(n, desc)
    bytes (u8 tensor) when that is non-zero (split-K launch)"""
    with _timed(_dkey("fwd_t", n, desc)):
        _check(load().sf_conv_fwd_t(_raw_in(inp, desc), i64(in_sample_stride), ptr(wt, "f32", "wt"),
                                    ptr(bias, "f32", "bias"), ptr(out, "f32", "out"), i64(n), C.byref(desc),
                                    ptr(workspace, "u8", "workspace"),
                                    i64(workspace.numel() if workspace is not None else 0), stream()),
               "sf_conv_fwd_t")


def tanh_scale_fwd(x, ld, n, col0, ncols, scale) -> None:
    _check(load().sf_tanh_scale_fwd(ptr(x, "f32", "x"), int(ld), i64(n), int(col0), int(ncols), C.c_float(scale), stream()),
           "sf_tanh_scale_fwd")


def tanh_scale_bwd(g, y, ld, n, col0, ncols, scale) -> None:
    _check(load().sf_tanh_scale_bwd(ptr(g, "f32", "g"), ptr(y, "f32", "y"), int(ld), i64(n), int(col0), int(ncols),
                                    C.c_float(scale), stream()), "sf_tanh_scale_bwd")


def transpose(w, wt, K, N) -> None:
    _check(load().sf_transpose(ptr(w, "f32", "w"), ptr(wt, "f32", "wt"), int(K), int(N), stream()), "sf_transpose")


def linear_fwd(inp, w, bias, out, M, K, N, relu) -> None:
    _check(load().sf_linear_fwd(ptr(inp, "f32", "in"), ptr(w, "f32", "w"), ptr(bias, "f32", "bias"),
                                ptr(out, "f32", "out"), i64(M), int(K), int(N), int(bool(relu)), stream()),
           "sf_linear_fwd")


def linear_wgrad_workspace(M, K, N) -> int:
    return int(load().sf_linear_wgrad_workspace(i64(M), int(K), int(N)))


def linear_wgrad(inp, dout, dw, db, M, K, N, workspace) -> None:
    _check(load().sf_linear_wgrad(ptr(inp, "f32", "in"), ptr(dout, "f32", "dout"), ptr(dw, "f32", "dw"),
                                  ptr(db, "f32", "db"), i64(M), int(K), int(N), ptr(workspace, "u8", "workspace"),
                                  stream()), "sf_linear_wgrad")


def linear_dgrad(dout, w, in_act, din, M, K, N) -> None:
    _check(load().sf_linear_dgrad(ptr(dout, "f32", "dout"), ptr(w, "f32", "w"), ptr(in_act, "f32", "in_act"),
                                  ptr(din, "f32", "din"), i64(M), int(K), int(N), stream()), "sf_linear_dgrad")


def relu_mask(g, act) -> None:
    _check(load().sf_relu_mask(ptr(g, "f32", "g"), ptr(act, "f32", "act"), i64(g.numel()), stream()), "sf_relu_mask")


conv_fwd = conv_fwd_raw
conv_wgrad = conv_wgrad_raw


# ---- data-parallel replicas: the RCCL exchange behind the C-ABI (csrc/sf_dp.hip) ------------------------------------
DP_UNIQUE_ID_BYTES = 128


def dp_unique_id() -> bytes:
    """rank 0: a fresh communicator id, to be shipped to the other ranks over any host channel"""
    buf = C.create_string_buffer(DP_UNIQUE_ID_BYTES)
    _check(load().sf_dp_unique_id(buf), "sf_dp_unique_id")
    return buf.raw


def dp_comm_create(unique_id: bytes, nranks: int, rank: int) -> C.c_void_p:
    """collective over the ranks; the communicator lives on the CURRENT device of the calling rank"""
    if len(unique_id) != DP_UNIQUE_ID_BYTES:
        raise SfHipError(f"dp_comm_create: the id must be {DP_UNIQUE_ID_BYTES} bytes, got {len(unique_id)}")
    comm = C.c_void_p()
    _check(load().sf_dp_comm_create(C.c_char_p(unique_id), int(nranks), int(rank), C.byref(comm)), "sf_dp_comm_create")
    return comm


def dp_comm_destroy(comm) -> None:
    _check(load().sf_dp_comm_destroy(comm), "sf_dp_comm_destroy")


def dp_comm_info(comm):
    n, r = C.c_int(), C.c_int()
    _check(load().sf_dp_comm_info(comm, C.byref(n), C.byref(r)), "sf_dp_comm_info")
    return n.value, r.value


def _cur_or(stream_) -> C.c_void_p:
    return stream() if stream_ is None else C.c_void_p(stream_.cuda_stream)


def allreduce_grads(comm, grads: torch.Tensor, stream_=None) -> None:
    """in-place SUM over the replicas of a contiguous fp32 device tensor (a slice of the flat gradient), enqueued on
    `stream_` (default: the current stream)"""
    _check(load().sf_allreduce_grads(comm, ptr(grads, "f32", "grads"), i64(grads.numel()), _cur_or(stream_)),
           "sf_allreduce_grads")


def dp_allreduce_f64(comm, buf: torch.Tensor, op: str = "sum", stream_=None) -> None:
    _check(load().sf_dp_allreduce_f64(comm, ptr(buf, "f64", "buf"), i64(buf.numel()), {"sum": 0, "max": 1}[op],
                                      _cur_or(stream_)), "sf_dp_allreduce_f64")


def dp_broadcast(comm, buf: torch.Tensor, root: int = 0, stream_=None) -> None:
    if not (buf.is_cuda and buf.is_contiguous()):
        raise SfHipError("dp_broadcast: a contiguous device tensor is required")
    _check(load().sf_dp_broadcast(comm, C.c_void_p(buf.data_ptr()), i64(buf.numel() * buf.element_size()), int(root),
                                  _cur_or(stream_)), "sf_dp_broadcast")


# ---- one-shot small-bucket exchange (csrc/sf_dp.hip, include/sf_hip.h: mailboxes mapped over hipIpc, one kernel per call)
DP_IPC_HANDLE_BYTES = 64


def dp_oneshot_create(nranks: int, rank: int, max_bytes: int):
    """-> (context, this rank's mailbox handle: DP_IPC_HANDLE_BYTES bytes to ship to every rank)"""
    ctx, h = C.c_void_p(), C.create_string_buffer(DP_IPC_HANDLE_BYTES)
    _check(load().sf_dp_oneshot_create(int(nranks), int(rank), i64(max_bytes), C.byref(ctx), h), "sf_dp_oneshot_create")
    return ctx, bytes(h.raw)


def dp_oneshot_connect(ctx, all_handles: bytes) -> None:
    """all_handles: the handles of ALL ranks, rank order, concatenated"""
    _check(load().sf_dp_oneshot_connect(ctx, C.c_char_p(all_handles)), "sf_dp_oneshot_connect")


def dp_oneshot_allreduce(ctx, buf: torch.Tensor, op: str = "sum", stream_=None) -> None:
    """in-place all-reduce of a contiguous f32 (sum) or f64 (sum / max) device tensor that fits the mailbox"""
    if buf.dtype == torch.float32:
        if op != "sum":
            raise SfHipError("dp_oneshot_allreduce: f32 buckets are summed")
        _check(load().sf_dp_oneshot_allreduce_f32(ctx, ptr(buf, "f32", "buf"), i64(buf.numel()), _cur_or(stream_)),
               "sf_dp_oneshot_allreduce_f32")
    else:
        _check(load().sf_dp_oneshot_allreduce_f64(ctx, ptr(buf, "f64", "buf"), i64(buf.numel()), {"sum": 0, "max": 1}[op],
                                                  _cur_or(stream_)), "sf_dp_oneshot_allreduce_f64")


def dp_oneshot_status(ctx) -> None:
    _check(load().sf_dp_oneshot_status(ctx), "sf_dp_oneshot_status")


def dp_oneshot_destroy(ctx) -> None:
    _check(load().sf_dp_oneshot_destroy(ctx), "sf_dp_oneshot_destroy")
